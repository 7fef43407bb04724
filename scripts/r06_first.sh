#!/bin/bash
# round 6, first GPU contact: the new idle-rank cases of the peer transport, the rows-to-root CRC through the multirank bench test, the latency
# sampler, then the bench line in the driver's form
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_dist_frame_loop.py -m gpu -q -x -k "6-peer or 7-peer or L24 or coherence" --durations=5 > gpurun_out/r06/idle_ranks.log 2>&1; echo "idle-rank tests rc=$?"; tail -8 gpurun_out/r06/idle_ranks.log
timeout 1500 python -m pytest tests/test_bench_multirank.py -m gpu -q -x > gpurun_out/r06/multirank.log 2>&1; echo "multirank bench rc=$?"; tail -5 gpurun_out/r06/multirank.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 2>gpurun_out/r06/bench_err.log | tee gpurun_out/r06/bench_driver_form.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['metric'], j['value'], j['ms_per_step'], 'steady', j['steady_state_ms_per_step'], 'latency', j['latency_ms'], j['latency_ms_max'], j['latency_samples'], j['roofline']['frac'], j['roofline']['hbm_frac'], j['cpu_baseline']['value'], j['kernel_ms'])"
