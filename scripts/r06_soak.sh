# Round 6 soak of the multi-GPU loop on this round's transport (plain light push + receiver-side rebuild, acquire behind every folded wait, the
# composite storing its own rows, the culls' deferred wait): long moving-camera sequences on ranks sharing the GPU, bit for bit against the
# single-GPU run at every checkpoint.  One gpurun call, from the repo root.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MV_TEST_TRANSPORT=peer MV_TEST_FOUR_CHAINS=1
out=gpurun_out/r06_soak; mkdir -p $out
{
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=4 timeout 900 python scripts/soak_dist.py 3 600 7 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=4 timeout 900 python scripts/soak_dist.py 2 600 64 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=3 timeout 900 python scripts/soak_dist.py 5 400 50 chains
MV_TEST_PEER_FOLD=1 timeout 900 python scripts/soak_dist.py 3 600 29 chains
MV_TEST_PEER_FOLD=0 timeout 900 python scripts/soak_dist.py 4 600 31 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=2 MV_TEST_PUSH_CELLS=1 timeout 900 python scripts/soak_dist.py 3 400 11 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=4 MV_TEST_ROWS_TO_ROOT=1 MV_TEST_W=250 MV_TEST_H=173 timeout 900 python scripts/soak_dist.py 4 300 13 chains
} 2>&1 | grep -v "^W\|warn" | tee $out/soak.txt
MV_TEST_CONFIG=C3 MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=4 OMP_NUM_THREADS=4 timeout 1200 python scripts/soak_dist.py 1 400 37 chains 2>&1 | grep -v "^W\|warn" | tee -a $out/soak.txt
# round 6: ranks that own no light layer (T2's L = 32 on 6 and 7 ranks, L = 24 on 5), and the single-GPU orbit soak with four chains
{
MV_TEST_PEER_FOLD=0 timeout 900 python scripts/soak_dist.py 6 300 17 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=2 timeout 900 python scripts/soak_dist.py 7 300 23 chains
MV_TEST_PEER_FOLD=1 MV_TEST_FRAMES_PER_LAUNCH=4 MV_TEST_L=24 MV_TEST_ROWS_TO_ROOT=1 timeout 900 python scripts/soak_dist.py 5 400 19 chains
} 2>&1 | grep -v "^W\|warn" | tee -a $out/soak.txt
MV_SOAK_FOUR_CHAINS=1 timeout 1500 python scripts/soak.py 20000 500 2>&1 | tail -5 | tee -a $out/soak.txt
