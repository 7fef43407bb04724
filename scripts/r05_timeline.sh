#!/bin/bash
# One gpurun call: the tile march's per-wavefront timeline (scripts/tile_timeline.py) for the product kernel and the register-diet experiment.
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/variants gpurun_out/r05
for v in ${VARIANTS:-"product:" "diet:-DMV_TILE_DIET"}; do
  name="${v%%:*}"; flags="${v#*:}"
  MV_OUT=/root/repo/gpurun_out/variants/tl_$name.so bash multivolumes_amd/csrc/build.sh -DMV_TILE_TIMELINE $flags > gpurun_out/variants/tl_$name.log 2>&1 || { echo "$name: build failed"; tail -5 gpurun_out/variants/tl_$name.log; continue; }
  for cfg in ${CONFIGS:-C3}; do
    echo "== $name $cfg"
    MV_LIB_PATH=/root/repo/gpurun_out/variants/tl_$name.so timeout 600 python scripts/tile_timeline.py $cfg 2>&1 | tail -24
  done
done | tee gpurun_out/r05/tile_timeline.txt
rm -f gpurun_out/variants/*.so
