cd /root/repo; mkdir -p gpurun_out/variants
for v in "nofloat:-DMV_DIET_FLOATS=0" "noptr:-DMV_DIET_PTRS=0" "none:-DMV_DIET_PTRS=0 -DMV_DIET_FLOATS=0"; do
  name="${v%%:*}"; flags="${v#*:}"
  MV_OUT=/root/repo/gpurun_out/variants/$name.so bash multivolumes_amd/csrc/build.sh $flags > gpurun_out/variants/$name.log 2>&1
  echo "== $name"; MV_LIB_PATH=/root/repo/gpurun_out/variants/$name.so python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k large_grid 2>&1 | tail -2
done
rm -f gpurun_out/variants/*.so
