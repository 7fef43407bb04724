for w in 5 6 7 8; do
  ./multivolumes_amd/csrc/build.sh -DMV_MARCH_MIN_WAVES=$w > /dev/null 2>&1
  for a in "" "--stress"; do
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves $w $a', round(d['value']), d['kernel_ms']['march_view'], d['roofline']['frac'])"
  done
done
