#!/bin/bash
# every documented build switch must still build and keep parity: build each into a library of its own, run the core parity tests on it
# (through gpurun, from the repo root; the round-3 experiments of the tile march -- MV_TILE_STAGE / _PRIORITY / _SPECULATE_BELOW -- build
# csrc/experiments/mv_march_tiles_experiments.hip in place of the product kernel: csrc/build.sh)
mkdir -p gpurun_out/variants
for f in -DMV_OCCUPANCY=0 -DMV_ALPHA_QUADS=0 -DMV_LIGHT_SPECULATE=0 -DMV_LIGHT_SPECULATE=2 -DMV_PREDICT_LIT=0 -DMV_TILE_WAVES=4 -DMV_TILE_LUT=0 -DMV_BRICK_PAD=0 -DMV_DIAGNOSTICS "-DMV_REFILL_MIN=1" "-DMV_REFILL_MIN=32" -DMV_ALPHA_QUADS=2 -DMV_LIGHT_LUT=0 -DMV_LIGHT_ORDER=0 -DMV_LIGHT_WAVES=2 -DMV_ORDER_DILATE=0 -DMV_ORDER_DILATE=2 -DMV_COMPOSITE_MIN_WAVES=6 -DMV_COMPOSITE_LDS=8192 -DMV_TILE_STAGE=1 -DMV_TILE_PRIORITY=1 -DMV_TILE_SPECULATE_BELOW=100000000 -DMV_LIGHT_CANDIDATES=1 -DMV_LIGHT_FMA_MIX=0 -DMV_STEP_RCP=0 -DMV_LIGHT_LAYERS_INTERLEAVED=1 -DMV_TILE_NARROW_BELOW=100000000; do
  name=$(echo "$f" | tr -c 'A-Za-z0-9' '_')
  skip=""; [ "$f" = "-DMV_OCCUPANCY=0" ] && skip=" and not headline"   # (that test also asserts that the occupancy bits skipped fetches)
  if MV_OUT=$PWD/gpurun_out/variants/$name.so ./multivolumes_amd/csrc/build.sh $f > /tmp/build.log 2>&1; then
    r=$(MV_LIB_PATH=$PWD/gpurun_out/variants/$name.so python -m pytest tests/test_gpu_parity.py tests/test_occupancy.py tests/test_gpu_configs.py -m gpu -q -x -k "(pipeline_parity or composite_parity or uniform_alpha or headline or tile_count_jumps or light_march_block_order or skip_the_view_march)$skip" 2>&1 | tail -1)
    echo "$f : $r"
  else
    echo "$f : BUILD FAILED"; grep error /tmp/build.log | head -3
  fi
done
rm -f gpurun_out/variants/*.so
