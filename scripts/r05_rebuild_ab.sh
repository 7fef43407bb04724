#!/bin/bash
# A/B of build switches on a rank of eight (loopback): scripts/r05_rebuild_ab.sh "name:-DFLAGS" ...  (MODES, default "peer4x4 peer4x4-root")
cd /root/repo; export HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p gpurun_out/ab
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"; mkdir -p gpurun_out/ab/$name
  MV_OUT=/root/repo/gpurun_out/ab/$name/libmultivolumes_amd.so bash multivolumes_amd/csrc/build.sh $flags > gpurun_out/ab/$name.log 2>&1 || { echo "$name: build failed"; continue; }
done
for rep in 1 2; do for v in "$@"; do name="${v%%:*}"; for m in ${MODES:-peer4x4 peer4x4-root}; do
  echo "$name $m: plain $(LD_LIBRARY_PATH=/root/repo/gpurun_out/ab/$name ./examples/dist_rank_cost.bin 400 8 -1 0 C3 $m 2>&1 | grep -o "max [0-9.]*") cells $(RANK_COST_PUSH_CELLS=1 LD_LIBRARY_PATH=/root/repo/gpurun_out/ab/$name ./examples/dist_rank_cost.bin 400 8 -1 0 C3 $m 2>&1 | grep -o "max [0-9.]*")"
done; done; done
rm -rf gpurun_out/ab/*/
