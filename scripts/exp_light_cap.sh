#!/bin/bash
# light march: requests in flight per ray (MV_LIGHT_SPECULATE) x occupancy cap through an unused LDS allocation (MV_LIGHT_LDS bytes per block)
for k in "$@"; do
  ./multivolumes_amd/csrc/build.sh -DMV_LIGHT_SPECULATE=$k > /dev/null 2>&1 || { echo "build failed: $k"; continue; }
  for l in 0 16384 24576 32768; do
    MV_LIGHT_LDS=$l python bench.py --no-cpu-baseline --steps 600 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$k lds=$l', round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms_in_flight'].items()}, round(d['kernel_ms']['march_light'],4))"
  done
done
./multivolumes_amd/csrc/build.sh > /dev/null 2>&1
