# where the host's time goes per mv_dist_frame call with several frames per launch (-DMV_HOST_PROFILE build beside the product library)
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r04_batch; mkdir -p $out/prof
MV_OUT=$PWD/$out/prof/libmultivolumes_amd.so ./multivolumes_amd/csrc/build.sh -DMV_HOST_PROFILE > /tmp/build_prof.log 2>&1 || tail -5 /tmp/build_prof.log
for t in ${MODES:-peer4 peer4x4}; do
LD_LIBRARY_PATH=$PWD/$out/prof:$LD_LIBRARY_PATH timeout 300 ./examples/dist_rank_cost.bin 400 8 2 0 C3 $t 2>&1 | sed "s/^/C3 $t: /"
done | tee $out/host_profile.txt
rm -rf $out/prof
