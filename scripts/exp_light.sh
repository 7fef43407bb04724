for w in 0 1; do
  ./multivolumes_amd/csrc/build.sh -DMV_LIGHT_SPECULATE=$w > /dev/null 2>&1
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('light speculate $w', d['kernel_ms'], d['ms_per_step'])"
done
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
