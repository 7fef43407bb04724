#!/bin/bash
# usage: scripts/r05_pmc_compare.sh "name:-DFLAGS" ... -- COUNTER ...   : the view march's counters per variant build (one --pmc pass per counter)
cd /root/repo; mkdir -p gpurun_out/variants
vars=(); while [ "$1" != "--" ]; do vars+=("$1"); shift; done; shift
for v in "${vars[@]}"; do
  name="${v%%:*}"; flags="${v#*:}"
  MV_OUT=/root/repo/gpurun_out/variants/$name.so bash multivolumes_amd/csrc/build.sh $flags > gpurun_out/variants/$name.log 2>&1 || { echo "$name: build failed"; continue; }
  echo "== $name"; MV_LIB_PATH=/root/repo/gpurun_out/variants/$name.so bash scripts/pmc.sh r05_$name "" "$@" 2>&1 | grep "marchTilesKernel<false"
done
rm -f gpurun_out/variants/*.so
