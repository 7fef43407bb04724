# usage: exp_flag.sh MACRO v1 v2 ...   -> rebuilds with -DMACRO=v and prints C3 / stress kernel times
M=$1; shift
for w in "$@"; do
  ./multivolumes_amd/csrc/build.sh -D$M=$w -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A3 "marchViewKernelILi0" | grep VGPRs | sed 's/.*remark: *//'
  for a in "" "--stress"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$M=$w $a', d['kernel_ms'], d['roofline']['frac'])"
  done
done
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
