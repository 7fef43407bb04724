# Round 6, last soak on the final sources: the configurations of scripts/r06_soak.sh at ten times the frames (the multi-GPU loop on ranks sharing the GPU, bit for
# bit against the single-GPU run at every checkpoint), then 200 000 frames of the single-GPU orbit on the four-chain context.  One gpurun call, from the repo root.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MV_TEST_TRANSPORT=peer MV_TEST_FOUR_CHAINS=1
out=gpurun_out/r06_soak_long; mkdir -p $out
{
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=4 timeout 1500 python scripts/soak_dist.py 3 6000 97 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=3 timeout 1500 python scripts/soak_dist.py 5 4000 211 chains
MV_TEST_PEER_FOLD=0 timeout 1500 python scripts/soak_dist.py 4 6000 331 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=2 MV_TEST_PUSH_CELLS=1 timeout 1500 python scripts/soak_dist.py 3 4000 113 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=4 MV_TEST_ROWS_TO_ROOT=1 MV_TEST_W=250 MV_TEST_H=173 timeout 1500 python scripts/soak_dist.py 8 3000 131 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=2 timeout 1500 python scripts/soak_dist.py 7 3000 233 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=4 MV_TEST_L=24 MV_TEST_ROWS_TO_ROOT=1 timeout 1500 python scripts/soak_dist.py 5 4000 191 chains
} 2>&1 | grep -v "^W\|warn" | tee $out/soak.txt
MV_TEST_CONFIG=C3 OMP_NUM_THREADS=4 timeout 1500 python scripts/soak_dist.py 2 2000 173 chains 2>&1 | grep -v "^W\|warn" | tee -a $out/soak.txt
MV_SOAK_FOUR_CHAINS=1 timeout 1500 python scripts/soak.py 200000 10000 2>&1 | tail -8 | tee -a $out/soak.txt
