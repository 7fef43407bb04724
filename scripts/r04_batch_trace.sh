# kernel timeline of one rank of eight with several frames per launch (run from the repo root, one gpurun call)
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r04_batch; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for t in ${MODES:-peer4x4}; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$t -- ./examples/dist_rank_cost.bin 400 8 ${RANK:-2} 0 C3 $t > $out/trace_$t.log 2>&1
python3 scripts/timeline.py $out/trace_$t > $out/rank_timeline_$t.txt 2>&1
for f in $(find $out/trace_$t -name "*kernel_trace.csv"); do cut -d, -f8-12 $f | tail -4000 > $out/kernel_trace_$t.csv; head -1 $f > $out/kernel_trace_header.txt; done
rm -rf $out/trace_$t
done
