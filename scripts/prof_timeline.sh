#!/bin/bash
# kernel-trace timeline of the default (frames in flight) bench: which kernels overlap, idle gaps (run through gpurun)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python3 /root/repo/bench.py --no-cpu-baseline --steps 300 --warmup 50 --roofline-frames 0 --steady-frames 0 --latency-frames 0 "$@" > /dev/null 2>&1
python3 /root/repo/scripts/timeline.py /tmp/prof_tl
