cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -- python3 /root/repo/bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-pipeline --roofline-frames 0 > /dev/null 2>&1
f=$(ls /tmp/prof_g/*/*kernel_trace.csv | head -1)
head -1 $f
grep marchView $f | tail -2
grep raySetup $f | tail -1
cd /root/repo
for s in 60 100 150 250; do export MV_MARCH_SHARE=$s; bash scripts/exp_variants.sh "--no-pipeline" "r64_share$s:-DMV_REFILL_MIN=64 -DMV_DIAGNOSTICS"; done
