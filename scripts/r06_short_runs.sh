#!/bin/bash
# round 6: what the driver's K = 20 timed frames (from an empty pipeline to a drained one) cost a rank of eight per frame, by frames per launch --
# examples/dist_rank_cost.bin <frames> warms up with min(200, frames) frames, synchronises, times <frames> frames + flush + synchronise (loopback, no fabric)
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r06_short; mkdir -p $out
for frames in ${FRAMES:-20 40 400}; do
for mode in ${MODES:-peer4-root peer4x2-root peer4x3-root peer4x4-root}; do
for rep in 1 2 3; do
timeout 300 ./examples/dist_rank_cost.bin $frames ${WORLDS:-8} ${RANK:-3} 0 C3 $mode 2>&1 | sed "s/^/frames $frames $mode: /"
done
done
done 2>&1 | tee $out/short.txt
