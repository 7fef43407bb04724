#!/bin/bash
# usage: scripts/pmc.sh <tag> "<bench args>" COUNTER [COUNTER ...]   -- one rocprofv3 --pmc pass per counter (never combined with API traces)
tag="$1"; args="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  out=/root/repo/gpurun_out/pmc_${tag}_${c}
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -- python3 /root/repo/bench.py --no-cpu-baseline --no-pipeline --steps 12 --warmup 4 --roofline-frames 0 --steady-frames 0 --latency-frames 0 $args > $out.log 2>&1
  python3 - "$out" "$c" <<'PY'
import csv, glob, sys, collections
d, c = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print(c, "no output"); sys.exit()
v = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == c: v[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for k, x in v.items():
    if k.startswith("mv::march") or k.startswith("mv::composite") or k.startswith("mv::cull") or k.startswith("mv::orders") or k.startswith("mv::light") or k.startswith("mv::exchange") or k.startswith("mv::raySetup") or k.startswith("mv::copyForward"):
        t = x[-8:]
        print("%-14s %-28s n %3d  mean(last 8) %.6g" % (c, k, len(x), sum(t) / len(t)))
PY
  rm -rf $out
done
