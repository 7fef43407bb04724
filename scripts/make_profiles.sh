#!/bin/bash
# Regenerates the evidence under profiles/ on an MI355X box (run through gpurun from the repo root). Round tag in $1 (default r01).
R=${1:-r01}
OUT=/root/repo/gpurun_out/profiles_$R
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
# 1. HBM traffic of every kernel: FETCH_SIZE / WRITE_SIZE in separate --pmc passes (no API tracing), C3 and C3-stress
bash scripts/pmc.sh c3 "" FETCH_SIZE WRITE_SIZE > $OUT/pmc_c3.txt 2>&1
bash scripts/pmc.sh c3s "--stress" FETCH_SIZE WRITE_SIZE > $OUT/pmc_c3_stress.txt 2>&1
python3 - $OUT $R <<'PY'
import json, re, sys
out, R = sys.argv[1], sys.argv[2]
res = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-pipeline --steps 12 --warmup 4`, mean of the last 8 launches, KB per launch; "
               "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE reports half the fetched bytes (MI355X_MICROARCH.md; calibrated on retileKernel: 134217728 B read -> 65564 KB)", "configs": {}}
traffic = {}
for tag, name in (("c3", "C3"), ("c3_stress", "C3-stress")):
    k = {}
    for line in open(f"{out}/pmc_{tag}.txt"):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+(\S+)\s+n\s+\d+\s+mean\(last 8\)\s+(\S+)", line)
        if m: k.setdefault(m.group(2), {})[m.group(1) + "_KB_per_launch"] = float(m.group(3))
    for v in k.values(): v["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
    res["configs"][name] = k
    mv = [v for n, v in k.items() if "marchViewKernel" in n]
    if mv: traffic[name] = {"marchViewKernel_hbm_bytes_per_launch": mv[0]["hbm_bytes_per_launch"]}
json.dump(res, open(f"{out}/{R}_pmc_hbm_traffic.json", "w"), indent=1)
json.dump(traffic, open("/root/repo/profiles/traffic.json", "w"), indent=1)   # bench.py reads it for roofline.traffic
json.dump(traffic, open(f"{out}/traffic.json", "w"), indent=1)
PY
# 2. kernel time: default (frames in flight) and --no-pipeline, under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
for mode in "" "--no-pipeline"; do
  tag=${mode:+_no_pipeline}
  rm -rf /tmp/prof$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$tag -- python3 /root/repo/bench.py --no-cpu-baseline $mode > $OUT/${R}_C3_bench${tag}_under_rocprofv3.json 2> $OUT/rocprof$tag.log
  cp /tmp/prof$tag/*/*kernel_stats.csv $OUT/${R}_C3_kernel_stats$tag.csv
done
cd /root/repo
# 3. un-profiled bench lines
python3 bench.py > $OUT/${R}_C3_bench.json 2> /dev/null
python3 bench.py --no-pipeline --no-cpu-baseline > $OUT/${R}_C3_bench_no_pipeline.json 2> /dev/null
python3 bench.py --stress --no-cpu-baseline > $OUT/${R}_C3_stress_bench.json 2> /dev/null
python3 bench.py --stress --no-pipeline --no-cpu-baseline > $OUT/${R}_C3_stress_bench_no_pipeline.json 2> /dev/null
for f in $OUT/*.json; do tail -1 $f > $f.tmp 2>/dev/null; done
for f in $OUT/${R}_C3*bench*.json; do mv $f.tmp $f; done; rm -f $OUT/*.tmp
grep -h marchView $OUT/${R}_C3_kernel_stats*.csv
for f in $OUT/${R}_C3*bench*.json; do python3 -c "import sys,json; d=json.loads(open('$f').read()); print('$f'.split('/')[-1], round(d['ms_per_step'],4), round(d['value']), d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline'].get('launch_ms_in_flight'))"; done
