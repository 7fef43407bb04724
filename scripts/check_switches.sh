#!/bin/bash
# every documented build switch must still build and keep parity: build each into a library of its own, run the core parity tests on it
# (through gpurun, from the repo root)
mkdir -p gpurun_out/variants
for f in -DMV_OCCUPANCY=0 -DMV_ALPHA_QUADS=0 -DMV_LIGHT_SPECULATE=0 -DMV_LIGHT_SPECULATE=2 -DMV_PREDICT_LIT=0 -DMV_TILE_WAVES=4 -DMV_TILE_LUT=0 -DMV_BRICK_PAD=0 -DMV_DIAGNOSTICS "-DMV_REFILL_MIN=1" "-DMV_REFILL_MIN=32"; do
  name=$(echo "$f" | tr -c 'A-Za-z0-9' '_')
  if MV_OUT=$PWD/gpurun_out/variants/$name.so ./multivolumes_amd/csrc/build.sh $f > /tmp/build.log 2>&1; then
    r=$(MV_LIB_PATH=$PWD/gpurun_out/variants/$name.so python -m pytest tests/test_gpu_parity.py tests/test_occupancy.py tests/test_gpu_configs.py -m gpu -q -x -k "pipeline_parity or composite_parity or uniform_alpha or headline or tile_count_jumps" 2>&1 | tail -1)
    echo "$f : $r"
  else
    echo "$f : BUILD FAILED"; grep error /tmp/build.log | head -3
  fi
done
rm -f gpurun_out/variants/*.so
