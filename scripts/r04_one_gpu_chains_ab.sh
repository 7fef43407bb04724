# one GPU: three chains + a stream for upload and cull (--cull-on-stream-a, the library's default) against four chains with the culls on them (bench.py's default since round 4)
for c in "" "--config C2" "--config C5 --steps 200" "--stress" "--orbit"; do
for i in 1 2; do
for a in "--cull-on-stream-a" ""; do
python bench.py --no-cpu-baseline --roofline-frames 0 $c $a 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('one GPU [$c] [$a]', round(j['ms_per_step'],4))"
done; done; done
