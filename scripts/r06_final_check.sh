#!/bin/bash
# what the driver runs at round end, in one gpurun call: the GPU suite, __graft_entry__.smoke(), the bench line in the driver's form
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r06t
timeout 3000 python -m pytest tests -m gpu -q -x --durations=30 > gpurun_out/r06t/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -14 gpurun_out/r06t/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/r06t/bench_driver_form.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['metric'], j['value'], j['ms_per_step'], 'steady', j['steady_state_ms_per_step'], 'latency', j['latency_ms'], j['roofline']['frac'], j['roofline']['hbm_frac'], j['cpu_baseline']['value'], j['frames_in_flight_total'])"
if [ -n "$SEEDS" ]; then MV_RANDOM_SEEDS=$SEEDS timeout 3000 python -m pytest tests/test_random_scenes.py -m gpu -q -x > gpurun_out/r06t/random_scenes.log 2>&1; echo "random scenes ($SEEDS seeds) rc=$?"; tail -3 gpurun_out/r06t/random_scenes.log; fi
