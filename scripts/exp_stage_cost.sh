cd /root/repo
for a in "" "--no-light" "--no-composite" "--no-light --no-composite"; do
  python bench.py --no-cpu-baseline --steps 600 $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$a', 'ms/frame', round(d['ms_per_step'],4), 'alone', {k:round(v,4) for k,v in d['kernel_ms'].items()}, 'inflight', {k:round(v,4) for k,v in d['kernel_ms_in_flight'].items()})"
done
