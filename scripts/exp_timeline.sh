cd /root/repo; mkdir -p gpurun_out/variants
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  MV_OUT=/root/repo/gpurun_out/variants/$name.so bash multivolumes_amd/csrc/build.sh -DMV_TIMELINE $flags > gpurun_out/variants/$name.log 2>&1 || { echo "$name: build failed"; tail -5 gpurun_out/variants/$name.log; continue; }
  echo "== $name"; MV_LIB_PATH=/root/repo/gpurun_out/variants/$name.so python scripts/march_timeline.py 2>&1 | tail -6
done
rm -f gpurun_out/variants/*.so
