#!/bin/bash
# usage: scripts/exp_flag.sh "<bench args>" -DFLAG=a -DFLAG=b ...   (rebuilds the library per flag set and prints the bench summary)
args="$1"; shift
for f in "$@"; do
  ./multivolumes_amd/csrc/build.sh $f > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  for i in 1 2; do
    python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
  done
done
./multivolumes_amd/csrc/build.sh > /dev/null 2>&1
