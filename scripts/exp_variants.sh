#!/bin/bash
# A/B timing of build switches: scripts/exp_variants.sh "<bench args>" "name1:-DFLAG=a -DFLAG2=b" "name2:..."  (run through gpurun)
# Every variant is built into its own shared object (gpurun_out/variants/<name>.so) and loaded through MV_LIB_PATH.
cd /root/repo
args="$1"; shift
mkdir -p gpurun_out/variants
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  MV_OUT=/root/repo/gpurun_out/variants/$name.so bash multivolumes_amd/csrc/build.sh $flags > gpurun_out/variants/$name.log 2>&1 || { echo "$name: build failed"; tail -5 gpurun_out/variants/$name.log; continue; }
  MV_LIB_PATH=/root/repo/gpurun_out/variants/$name.so python bench.py --no-cpu-baseline --steps 400 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('%-28s ms/frame %.4f  marchV alone %.4f inflight %.4f  light %.4f comp %.4f  lane_util %.3f' % ('$name', d['ms_per_step'], d['kernel_ms']['march_view'], d['kernel_ms_in_flight']['march_view'], d['kernel_ms']['march_light'], d['kernel_ms']['composite'], c['lane_utilisation'] or 0))"
done
rm -f gpurun_out/variants/*.so
