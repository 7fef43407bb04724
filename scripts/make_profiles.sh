#!/bin/bash
# Regenerates the evidence under profiles/ on an MI355X box (run through gpurun from the repo root). Round tag in $1 (default r02).
R=${1:-r06}
OUT=/root/repo/gpurun_out/profiles_$R
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
HASH=$(python3 -c "import bench; print(bench.march_source_hash())")
# 1. HBM traffic of every kernel: FETCH_SIZE / WRITE_SIZE in separate --pmc passes (no API tracing), C3 and C3-stress
bash scripts/pmc.sh c3 "" FETCH_SIZE WRITE_SIZE > $OUT/pmc_c3.txt 2>&1
bash scripts/pmc.sh c3s "--stress" FETCH_SIZE WRITE_SIZE > $OUT/pmc_c3_stress.txt 2>&1
bash scripts/pmc.sh c2 "--config C2" FETCH_SIZE WRITE_SIZE > $OUT/pmc_c2.txt 2>&1
bash scripts/pmc.sh c5 "--config C5" FETCH_SIZE WRITE_SIZE > $OUT/pmc_c5.txt 2>&1
python3 - $OUT $R $HASH <<'PY'
import json, re, sys
out, R, H = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-pipeline --steps 12 --warmup 4`, mean of the last 8 launches, KB per launch; "
               "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE reports half the fetched bytes (MI355X_MICROARCH.md; calibrated on retileKernel: 134217728 B read -> 65564 KB)",
       "source_hash": H, "configs": {}}
traffic = {"source_hash": H}
for tag, name in (("c3", "C3"), ("c3_stress", "C3-stress"), ("c2", "C2"), ("c5", "C5")):
    k = {}
    for line in open(f"{out}/pmc_{tag}.txt"):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+(.+?)\s+n\s+\d+\s+mean\(last 8\)\s+(\S+)", line)
        if m: k.setdefault(m.group(2), {})[m.group(1) + "_KB_per_launch"] = float(m.group(3))
    for v in k.values(): v["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
    res["configs"][name] = k
    mv = [v for n, v in k.items() if "marchTilesKernel<false" in n]
    if mv: traffic[name] = {"marchViewKernel_hbm_bytes_per_launch": mv[0]["hbm_bytes_per_launch"]}
json.dump(res, open(f"{out}/{R}_pmc_hbm_traffic.json", "w"), indent=1)
json.dump(traffic, open("/root/repo/profiles/traffic.json", "w"), indent=1)   # bench.py reads it for roofline.traffic (only while source_hash matches)
json.dump(traffic, open(f"{out}/traffic.json", "w"), indent=1)
PY
# 2. VALU / L2 counters of every kernel (one --pmc pass per counter)
bash scripts/pmc.sh valu "" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES TA_BUSY_avr GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum > $OUT/${R}_pmc_valu_l2.txt 2>&1
# 3. kernel time: default (frames in flight) and --no-pipeline, under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
for mode in "" "--no-pipeline"; do
  tag=${mode:+_no_pipeline}
  rm -rf /tmp/prof$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$tag -- python3 /root/repo/bench.py --no-cpu-baseline --steady-frames 0 --latency-frames 0 $mode > $OUT/${R}_C3_bench${tag}_under_rocprofv3.json 2> $OUT/rocprof$tag.log
  cp /tmp/prof$tag/*/*kernel_stats.csv $OUT/${R}_C3_kernel_stats$tag.csv
done
cd /root/repo
# 4. un-profiled bench lines: headline, one stream, stress, moving camera, the reference scene's mesh, C5 and C2 on one GPU, the ray-queue march
python3 bench.py > $OUT/${R}_C3_bench.json 2> /dev/null
python3 bench.py --no-pipeline --no-cpu-baseline > $OUT/${R}_C3_bench_no_pipeline.json 2> /dev/null
python3 bench.py --stress --no-cpu-baseline > $OUT/${R}_C3_stress_bench.json 2> /dev/null
python3 bench.py --stress --no-pipeline --no-cpu-baseline > $OUT/${R}_C3_stress_bench_no_pipeline.json 2> /dev/null
python3 bench.py --orbit --no-cpu-baseline > $OUT/${R}_C3_orbit_bench.json 2> /dev/null
python3 bench.py --mesh --no-cpu-baseline > $OUT/${R}_C3_mesh_bench.json 2> /dev/null
python3 bench.py --mesh --orbit --no-cpu-baseline > $OUT/${R}_C3_mesh_orbit_bench.json 2> /dev/null
python3 bench.py --config C5 --steps 200 --warmup 40 --no-cpu-baseline > $OUT/${R}_C5_bench.json 2> /dev/null
python3 bench.py --config C2 --no-cpu-baseline > $OUT/${R}_C2_bench.json 2> /dev/null
python3 bench.py --ray-queue --no-cpu-baseline > $OUT/${R}_C3_rayqueue_bench.json 2> /dev/null
python3 bench.py --ray-queue --config C5 --steps 200 --warmup 40 --no-cpu-baseline > $OUT/${R}_C5_rayqueue_bench.json 2> /dev/null
for f in $OUT/*bench*.json; do tail -1 $f > $f.tmp 2>/dev/null && mv $f.tmp $f; done
bash scripts/exp_stage_cost.sh > $OUT/${R}_stage_cost.txt 2>&1
bash scripts/prof_timeline.sh > $OUT/${R}_timeline_pipelined.txt 2>&1
bash scripts/prof_timeline.sh --config C2 > $OUT/${R}_C2_timeline_pipelined.txt 2>&1
grep -h "march" $OUT/${R}_C3_kernel_stats*.csv | cut -d, -f1-4
for f in $OUT/*bench*.json; do python3 -c "import sys,json; d=json.loads(open('$f').read()); r=d['roofline']; print('$f'.split('/')[-1], round(d['ms_per_step'],4), round(d['value']), 'frac', round(r['frac'],3), 'with fused RayCast', round(r['frac_with_fused_raycast'],3), 'hbm_frac', r['hbm_frac'] and round(r['hbm_frac'],3), 'launch', round(r['launch_ms'],4), 'in flight', round(r['launch_ms_in_flight'],4))"; done
