# A/B of conv tile choices inside the step (conv_igemm.hip choose_tile hooks), same box, three rounds
run() { env $1 python bench.py --workload all --no-cpu-baseline --no-roofline-top --no-parity --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s' % '$1', d['ms_per_step'], d['fwd']['ms_per_step'])"; }
for rep in 1 2 3; do
run "HDRSKY_X=default"
run "HDRSKY_TILE_W256=1,8,4,1,32,1"
run "HDRSKY_TILE_W256=1,8,2,1,16,1"
done
