# A/B of build switches on a rank of eight with four frames per launch: every variant a library of its own, loaded through LD_LIBRARY_PATH
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r04_batch; mkdir -p $out
for v in "$@"; do
name=${v%%:*}; flags=${v#*:}
mkdir -p $out/lib_$name
MV_OUT=$PWD/$out/lib_$name/libmultivolumes_amd.so ./multivolumes_amd/csrc/build.sh $flags > /tmp/build_$name.log 2>&1 || tail -5 /tmp/build_$name.log
for rep in 1 2; do
LD_LIBRARY_PATH=$PWD/$out/lib_$name:$LD_LIBRARY_PATH timeout 300 ./examples/dist_rank_cost.bin 400 8 -1 0 C3 ${MODE:-peer4x4} 2>&1 | sed "s/^/$name: /"
done
rm -rf $out/lib_$name
done | tee $out/ab.txt
