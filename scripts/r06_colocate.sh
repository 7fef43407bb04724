#!/bin/bash
# round 6: several frames per launch on ONE GPU (world 1, peer loopback) -- does putting the same tile pair of every frame of a launch on the same
# XCD turn the frames' common texel reads into L2 hits?  A/B of -DMV_BATCH_COLOCATE, with the HBM traffic of the view-march launch (PMC).
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06_colocate; mkdir -p $out
for v in "$@"; do
name=${v%%:*}; flags=${v#*:}
mkdir -p $out/lib_$name
MV_OUT=$PWD/$out/lib_$name/libmultivolumes_amd.so ./multivolumes_amd/csrc/build.sh $flags > /tmp/build_$name.log 2>&1 || tail -5 /tmp/build_$name.log
for mode in ${MODES:-peer4 peer4x2 peer4x4}; do
for rep in 1 2; do
LD_LIBRARY_PATH=$PWD/$out/lib_$name:$LD_LIBRARY_PATH timeout 300 ./examples/dist_rank_cost.bin 600 1 -1 0 ${CONFIG:-C3} $mode 2>&1 | sed "s/^/$name $mode: /"
done
done
if [ "$PMC" != 0 ]; then
for c in FETCH_SIZE TCC_HIT_sum TCC_MISS_sum; do
  d=/tmp/pmc_$name_$c; rm -rf $d; mkdir -p $d
  (cd /tmp && TMPDIR=/tmp LD_LIBRARY_PATH=/root/repo/$out/lib_$name:$LD_LIBRARY_PATH rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- /root/repo/examples/dist_rank_cost.bin 48 1 -1 0 ${CONFIG:-C3} peer4x4 > $d.log 2>&1)
  python3 - "$d" "$c" "$name" <<'PY'
import csv, glob, sys, collections
d, c, name = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print(name, c, "no output"); sys.exit()
v = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == c: v[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for k, x in v.items():
    if "march" in k or "composite" in k:
        t = x[-6:]
        print("%s %-14s %-60s n %3d  mean(last 6) %.6g" % (name, c, k[:60], len(x), sum(t) / len(t)))
PY
done
fi
rm -rf $out/lib_$name
done 2>&1 | tee $out/ab.txt
