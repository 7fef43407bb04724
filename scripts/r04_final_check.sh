export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r04t
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r04t/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r04t/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['metric'], j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['hbm_frac'], j['cpu_baseline']['value'], j['config']['frames_in_flight'])"
