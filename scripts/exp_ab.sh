#!/bin/bash
# A/B of library builds under build/<name>/ (csrc/build.sh with MV_OUT=...): one-stream and pipelined frame, and the orbit.  usage: scripts/exp_ab.sh name [name ...]
for l in "$@"; do
  if [ "$l" = default ]; then unset MV_LIB_PATH; else export MV_LIB_PATH=/root/repo/build/$l/libmultivolumes_amd.so; fi
  for rep in 1 2; do
    a=$(python3 /root/repo/bench.py --no-cpu-baseline --no-pipeline --steps 300 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f view %.4f' % (d['ms_per_step'], d['kernel_ms']['march_view']))")
    b=$(python3 /root/repo/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f view alone %.4f' % (d['ms_per_step'], d['kernel_ms']['march_view']))")
    c=$(python3 /root/repo/bench.py --no-cpu-baseline --orbit 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f view alone %.4f' % (d['ms_per_step'], d['kernel_ms']['march_view']))")
    echo "$l: one stream $a | pipelined $b | orbit $c"
  done
done
