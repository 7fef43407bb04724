export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r04n
for nb in 0 128 2048; do
MV_OUT=$PWD/gpurun_out/r04n/nb$nb.so ./multivolumes_amd/csrc/build.sh -DMV_TILE_NARROW_BELOW=$nb > /tmp/b.log 2>&1 || tail -3 /tmp/b.log
for cfgn in T1 T2 C1 C2; do
MV_LIB_PATH=$PWD/gpurun_out/r04n/nb$nb.so python bench.py --config $cfgn --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('narrow below $nb', '$cfgn', 'frame ms', round(j['ms_per_step'],4), 'view march alone ms', round(j['roofline']['launch_ms'],4), 'rays', j['config']['rays_per_frame'])"
done
done
rm -f gpurun_out/r04n/*.so*
