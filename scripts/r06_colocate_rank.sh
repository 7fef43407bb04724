#!/bin/bash
# round 6: a rank of eight (loopback, no fabric), four frames per launch: do the frames of a launch that share frame 0's tile order on the same XCDs help a rank?
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06_colocate; mkdir -p $out
for v in "$@"; do
name=${v%%:*}; flags=${v#*:}
mkdir -p $out/lib_$name
MV_OUT=$PWD/$out/lib_$name/libmultivolumes_amd.so ./multivolumes_amd/csrc/build.sh $flags > /tmp/build_$name.log 2>&1 || tail -5 /tmp/build_$name.log
for mode in ${MODES:-peer4x4-root}; do
for rep in 1 2; do
LD_LIBRARY_PATH=$PWD/$out/lib_$name:$LD_LIBRARY_PATH timeout 600 ./examples/dist_rank_cost.bin 400 ${WORLDS:-8} -1 0 ${CONFIG:-C3} $mode 2>&1 | sed "s/^/$name $mode: /"
done
done
rm -rf $out/lib_$name
done 2>&1 | tee $out/rank.txt
