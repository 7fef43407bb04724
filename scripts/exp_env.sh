#!/bin/bash
# usage: scripts/exp_env.sh "<bench args>" "ENV=.. ENV2=.." ...
args="$1"; shift
for e in "$@"; do
  for i in 1 2; do
    env $e python bench.py --no-cpu-baseline --no-pipeline --steps 100 --warmup 30 $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
  done
done
