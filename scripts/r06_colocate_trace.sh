#!/bin/bash
# round 6: kernel timeline of world 1, four frames per launch, with and without the frames of a launch sharing frame 0's tile order on the same XCDs
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06_colocate; mkdir -p $out
for v in "$@"; do
name=${v%%:*}; flags=${v#*:}
mkdir -p $out/lib_$name
MV_OUT=$PWD/$out/lib_$name/libmultivolumes_amd.so ./multivolumes_amd/csrc/build.sh $flags > /tmp/build_$name.log 2>&1 || tail -5 /tmp/build_$name.log
for nocomp in 0 1; do
  if [ $nocomp = 1 ]; then export RANK_COST_NO_COMPOSITE=1; else unset RANK_COST_NO_COMPOSITE; fi
  LD_LIBRARY_PATH=$PWD/$out/lib_$name:$LD_LIBRARY_PATH timeout 300 ./examples/dist_rank_cost.bin 600 1 -1 0 C3 peer4x4 2>&1 | sed "s/^/$name nocomposite=$nocomp: /"
done
unset RANK_COST_NO_COMPOSITE
(cd /tmp && export TMPDIR=/tmp && LD_LIBRARY_PATH=/root/repo/$out/lib_$name:$LD_LIBRARY_PATH timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$out/trace_$name -- /root/repo/examples/dist_rank_cost.bin 400 1 -1 0 C3 peer4x4 > /root/repo/$out/trace_$name.log 2>&1)
python3 scripts/timeline.py $out/trace_$name 2>&1 | head -24 | sed "s/^/$name: /"
rm -rf $out/trace_$name $out/lib_$name
done 2>&1 | tee $out/trace.txt
