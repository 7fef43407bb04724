# soak of the multi-GPU loop with several frames per launch: long moving-camera sequences on ranks sharing the GPU, bit for bit against the single-GPU run at every checkpoint
export HSA_ENABLE_IPC_MODE_LEGACY=0 MV_TEST_TRANSPORT=peer MV_TEST_FOUR_CHAINS=1 MV_TEST_PEER_FOLD=1
out=gpurun_out/r04_batch; mkdir -p $out
{
MV_TEST_FRAMES_PER_LAUNCH=4 timeout 900 python scripts/soak_dist.py 3 600 7 chains
MV_TEST_FRAMES_PER_LAUNCH=4 timeout 900 python scripts/soak_dist.py 2 600 64 chains
MV_TEST_FRAMES_PER_LAUNCH=3 timeout 900 python scripts/soak_dist.py 5 400 50 chains
MV_TEST_FRAMES_PER_LAUNCH=2 MV_TEST_DEAL_VOLUMES=1 timeout 900 python scripts/soak_dist.py 3 400 11 chains
MV_TEST_FRAMES_PER_LAUNCH=4 MV_TEST_W=250 MV_TEST_H=173 timeout 900 python scripts/soak_dist.py 4 300 13 chains
} 2>&1 | grep -v "^W\|warn" | tee $out/soak.txt
# ... and at the full C3 frame on one rank (sixteen output sets of full size rotating): MV_TEST_CONFIG=C3
MV_TEST_CONFIG=C3 MV_TEST_FRAMES_PER_LAUNCH=4 OMP_NUM_THREADS=4 timeout 1200 python scripts/soak_dist.py 1 400 37 chains 2>&1 | grep -v "^W\|warn" | tee -a gpurun_out/r04_batch/soak.txt
