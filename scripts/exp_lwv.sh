for w in 0 1; do
  ./multivolumes_amd/csrc/build.sh -DMV_LIGHT_WITH_VOLUME=$w -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A3 "marchViewKernelILi0" | grep VGPRs
  for a in "" "--stress"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('light-with-volume $w $a', d['kernel_ms']['march_view'], d['roofline']['frac'])"
  done
done
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
