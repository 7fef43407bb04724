#!/bin/bash
# Round 5: what a rank of eight costs per C3 frame without a fabric (examples/dist_rank_cost.bin, loopback peer stores), for the modes in MODES,
# and the peer parity tests (TESTS = a -k expression, empty = none).  One gpurun call, from the repo root, after csrc/build.sh + examples/build.sh.
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r05_rank; mkdir -p $out
for t in ${MODES:-peer4 peer4x4 peer4x4-root}; do
  for rep in 1 ${REPS:-}; do timeout 300 ./examples/dist_rank_cost.bin 400 ${WORLDS:-9} -1 0 ${CONFIG:-C3} $t 2>&1 | sed "s/^/${CONFIG:-C3} $t: /"; done
done | tee $out/rank_cost.txt
if [ -n "$TESTS" ]; then timeout 2400 python -m pytest tests/test_dist_frame_loop.py -m gpu -x -q -k "$TESTS" --durations=5 > $out/tests.log 2>&1; echo "tests rc=$?"; tail -8 $out/tests.log; fi
