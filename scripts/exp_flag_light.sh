#!/bin/bash
for f in "$@"; do
  ./multivolumes_amd/csrc/build.sh $f > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  for a in "" "--config T2"; do
    python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f $a', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()}, d['config']['light_samples_per_frame'])"
  done
done
./multivolumes_amd/csrc/build.sh > /dev/null 2>&1
