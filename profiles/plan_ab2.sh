# A/B at 128x512: tiles of the <= 16- and 32-output-channel classes (hooks HDRSKY_TILE_C16 / HDRSKY_TILE_C32), same box
run() { env $1 python bench.py --workload $2 --no-cpu-baseline --no-roofline-top --no-parity --steps $3 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s %-60s' % ('$2', '$1'), d['ms_per_step'])"; }
for rep in 1 2; do
for wl in hires-train hires; do
run "HDRSKY_X=default" $wl 20
run "HDRSKY_TILE_C16=8,1,4,1,32,1" $wl 20
run "HDRSKY_TILE_C32=4,2,4,1,32,1" $wl 20
run "HDRSKY_TILE_C16=8,1,4,1,32,1 HDRSKY_TILE_C32=4,2,4,1,32,1" $wl 20
done
done
