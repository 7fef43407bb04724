# several frames per launch: what a rank costs, then parity on ranks sharing the GPU (one gpurun call, run from the repo root)
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r04_batch; mkdir -p $out
for t in ${MODES:-peer4 peer4x2 peer4x3 peer4x4 peer4x4-root}; do
timeout 300 ./examples/dist_rank_cost.bin 400 ${WORLDS:-9} -1 0 C3 $t 2>&1 | sed "s/^/C3 $t: /"
done | tee $out/rank_cost.txt
RANK_COST_DEAL_VOLUMES=1 timeout 300 ./examples/dist_rank_cost.bin 400 8 -1 0 C3 peer4 2>&1 | sed "s/^/C3 peer4 whole volumes: /" | tee -a $out/rank_cost.txt
timeout 1500 python -m pytest tests/test_dist_frame_loop.py -m gpu -x -q -k "${TESTS:-peer}" --durations=5 > $out/tests.log 2>&1; echo "tests rc=$?"; tail -15 $out/tests.log
