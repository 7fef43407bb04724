#!/bin/bash
# every documented build switch must still build and keep parity: rebuild with each, run the core parity tests
for f in -DMV_OCCUPANCY=0 -DMV_ALPHA_QUADS=0 -DMV_LIGHT_SPECULATE=0 -DMV_LIGHT_SPECULATE=2 -DMV_ALPHA_FROM_TEXELS=0 -DMV_PREDICT_LIT=0 -DMV_MARCH_WAVES=4 -DMV_BRICK_PAD=0; do
  if ./multivolumes_amd/csrc/build.sh $f > /tmp/build.log 2>&1; then
    r=$(python -m pytest tests/test_gpu_parity.py tests/test_occupancy.py -m gpu -q -x -k "pipeline_parity or composite_parity or uniform_alpha or headline" 2>&1 | tail -1)
    echo "$f : $r"
  else
    echo "$f : BUILD FAILED"; grep error /tmp/build.log | head -3
  fi
done
./multivolumes_amd/csrc/build.sh > /dev/null 2>&1
for e in MV_TILE_ORDER=0 MV_EXPERIMENT=0; do
  r=$(env $e python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipeline_parity or composite_parity" 2>&1 | tail -1)
  echo "$e : $r"
done
