mkdir -p gpurun_out/r04b
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_dist_frame_loop.py -m gpu -x -q -k "peer and not eight" > gpurun_out/r04b/peer_small.log 2>&1; echo "peer small rc=$?"
tail -8 gpurun_out/r04b/peer_small.log
for t in callbacks peer-nofold peer; do
  timeout 600 ./examples/dist_rank_cost.bin 400 9 -1 0 C3 $t > gpurun_out/r04b/rank_cost_$t.txt 2>&1; echo "$t rc=$?"; cat gpurun_out/r04b/rank_cost_$t.txt
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04b/trace_peer -- ./examples/dist_rank_cost.bin 400 8 2 0 C3 peer > gpurun_out/r04b/trace_peer.log 2>&1
python3 scripts/timeline.py gpurun_out/r04b/trace_peer > gpurun_out/r04b/rank_timeline_peer.txt 2>&1; head -40 gpurun_out/r04b/rank_timeline_peer.txt
rm -rf gpurun_out/r04b/trace_peer
# A/B of the tile march: the product kernel against round 3's (experiments file, switches at their defaults)
MV_OUT=$PWD/gpurun_out/r04b/old_tiles.so ./multivolumes_amd/csrc/build.sh -DMV_TILE_PRIORITY=0 > /tmp/build_old.log 2>&1 || tail -5 /tmp/build_old.log
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('new', j['ms_per_step'], j['roofline']['launch_ms'], j['kernel_ms'])"
  MV_LIB_PATH=$PWD/gpurun_out/r04b/old_tiles.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('old', j['ms_per_step'], j['roofline']['launch_ms'], j['kernel_ms'])"
done
rm -f gpurun_out/r04b/old_tiles.so*
