#!/bin/bash
# Round 5 A/B of build switches of the single-GPU frame, one gpurun call: every variant "name:-DFLAGS" is built into gpurun_out/variants/<name>.so,
# timed twice (alternating) pipelined at C3 (+ whatever BENCH_ARGS says), and its view march's VALU instructions counted (one --pmc pass).
# usage: scripts/r05_ab.sh "name:-DFLAG=a" ...      env: BENCH_ARGS (extra bench.py arguments), PMC=0 (skip the counter pass), TESTS="tests/x.py ..." (parity first, default library)
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/variants gpurun_out/r05
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -m gpu -q -x 2>&1 | tail -4; fi
names=()
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  MV_OUT=/root/repo/gpurun_out/variants/$name.so bash multivolumes_amd/csrc/build.sh $flags > gpurun_out/variants/$name.log 2>&1 || { echo "$name: build failed"; tail -5 gpurun_out/variants/$name.log; continue; }
  names+=("$name")
done
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; r=d['roofline']
print('%-22s ms/frame %.4f  marchV alone %.4f inflight %.4f  light %.4f comp %.4f  frac %.3f  samples %d skipped %d' % ('$1', d['ms_per_step'], d['kernel_ms']['march_view'], d['kernel_ms_in_flight']['march_view'], d['kernel_ms']['march_light'], d['kernel_ms']['composite'], r['frac'], c['view_samples_per_frame'], c['fetch_free_samples_per_frame']))"; }
for rep in 1 2; do
  for name in "${names[@]}"; do
    MV_LIB_PATH=/root/repo/gpurun_out/variants/$name.so python bench.py --no-cpu-baseline --steps 600 $BENCH_ARGS 2>/dev/null | line $name
  done
done
if [ "$PMC" != 0 ]; then
  for name in "${names[@]}"; do
    echo "== $name"; MV_LIB_PATH=/root/repo/gpurun_out/variants/$name.so bash scripts/pmc.sh r05_$name "$BENCH_ARGS" SQ_INSTS_VALU 2>&1 | grep -E "marchTiles|marchLight" 
  done
fi
rm -f gpurun_out/variants/*.so
