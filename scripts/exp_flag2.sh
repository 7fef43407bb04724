# usage: exp_flag2.sh MACRO v1 v2 ...   -> rebuilds with -DMACRO=v and prints C3 kernel times (all kernels)
M=$1; shift
for w in "$@"; do
  ./multivolumes_amd/csrc/build.sh -D$M=$w -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "marchLightKernel|compositeKernel" -A3 | grep VGPRs | sed 's/.*remark: *//' | tr '\n' ' '
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$M=$w', d['kernel_ms'], d['ms_per_step'])"
done
