# what a rank of the multi-GPU loop costs without a fabric (profiles/r04_peer_rank_cost.txt) and a rank's kernel timelines
# (profiles/r04_rank_timeline_world8*.txt): one gpurun call, run from the repo root after csrc/build.sh and the examples' build line
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r04_rank_cost; mkdir -p $out
for t in callbacks peer peer4 peer4-root; do
timeout 900 ./examples/dist_rank_cost.bin 400 15 -1 0 C3 $t 2>&1 | sed "s/^/C3 $t: /"
done | tee $out/rank_cost.txt
for t in callbacks peer4 peer4-root; do
timeout 900 ./examples/dist_rank_cost.bin 300 9 -1 0 C5 $t 2>&1 | sed "s/^/C5 $t: /"
done | tee -a $out/rank_cost.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for t in peer4 peer; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$t -- ./examples/dist_rank_cost.bin 400 8 2 0 C3 $t > $out/trace_$t.log 2>&1
python3 scripts/timeline.py $out/trace_$t > $out/rank_timeline_$t.txt 2>&1
rm -rf $out/trace_$t
done
