mkdir -p gpurun_out/r04j
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_dist_frame_loop.py -m gpu -q -k "peer and not eight and not side" 2>&1 | tail -3
for t in peer callbacks; do
  timeout 600 ./examples/dist_rank_cost.bin 400 8 -1 0 C3 $t 2>&1 | sed "s/^/$t: /"
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for t in peer; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04j/trace_$t -- ./examples/dist_rank_cost.bin 400 8 2 0 C3 $t > gpurun_out/r04j/trace_$t.log 2>&1
python3 scripts/timeline.py gpurun_out/r04j/trace_$t > gpurun_out/r04j/rank_timeline_$t.txt 2>&1; head -28 gpurun_out/r04j/rank_timeline_$t.txt | cut -c1-110
rm -rf gpurun_out/r04j/trace_$t
done
