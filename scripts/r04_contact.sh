export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_dist_frame_loop.py -m gpu -q -x -k "peer and not eight" 2>&1 | tail -3
for t in peer peer4 peer4-root; do
timeout 600 ./examples/dist_rank_cost.bin 400 8 -1 0 C3 $t 2>&1 | sed "s/^/C3 $t: /"
done
