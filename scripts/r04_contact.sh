mkdir -p gpurun_out/r04e
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
MV_OUT=$PWD/gpurun_out/r04e/plain.so ./multivolumes_amd/csrc/build.sh -DMV_PEER_PLAIN_STORES > /tmp/build_plain.log 2>&1 || tail -5 /tmp/build_plain.log
cp multivolumes_amd/libmultivolumes_amd.so gpurun_out/r04e/product.so
for v in product plain; do
cp gpurun_out/r04e/$v.so multivolumes_amd/libmultivolumes_amd.so
for t in peer peer-nofold; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04e/trace_$t -- ./examples/dist_rank_cost.bin 400 8 2 0 C3 $t > gpurun_out/r04e/trace_$t.log 2>&1
echo "== $v $t"; python3 scripts/timeline.py gpurun_out/r04e/trace_$t 2>&1 | head -22 | cut -c1-100
rm -rf gpurun_out/r04e/trace_$t
done
done
rm -f gpurun_out/r04e/*.so*
