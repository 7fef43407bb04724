#!/bin/bash
# per-kernel time under rocprofv3 (run through gpurun): scripts/prof_stats.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 /root/repo/bench.py --no-cpu-baseline --steps 200 --warmup 20 --steady-frames 0 --latency-frames 0 "$@" > /root/repo/gpurun_out/prof_$tag.json 2> /root/repo/gpurun_out/prof_$tag.log
cp /tmp/prof_$tag/*/*kernel_stats.csv /root/repo/gpurun_out/prof_${tag}_kernel_stats.csv
cut -d, -f1-8 /root/repo/gpurun_out/prof_${tag}_kernel_stats.csv | head -14
