mkdir -p gpurun_out/r04a
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_dist_frame_loop.py -m gpu -x -q -k "peer and not eight" > gpurun_out/r04a/peer_small.log 2>&1; echo "peer small rc=$?"
tail -5 gpurun_out/r04a/peer_small.log
timeout 600 ./examples/dist_rank_cost.bin 400 9 -1 0 C3 callbacks > gpurun_out/r04a/rank_cost_callbacks.txt 2>&1; echo "rc=$?"; cat gpurun_out/r04a/rank_cost_callbacks.txt
timeout 600 ./examples/dist_rank_cost.bin 400 9 -1 0 C3 peer > gpurun_out/r04a/rank_cost_peer.txt 2>&1; echo "rc=$?"; cat gpurun_out/r04a/rank_cost_peer.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?"; cat gpurun_out/r04a/bench.json
