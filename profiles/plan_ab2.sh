export HDRSKY_EXPERIMENTS=1   # the tuning hooks this script sets are behind the gate since round 4 (csrc/hooks.h, hooks.py)
# A/B helper of the session (last use: driver-line workload, narrow-output tile old vs new, same box)
run() { env $1 python bench.py --workload all --no-cpu-baseline --no-parity --no-roofline-top --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$1', d.get('ms_per_step'), d.get('fwd',{}).get('ms_per_step'))"; }
for rep in 1 2 3; do
run "HDRSKY_X=default"
run "HDRSKY_TILE_C16=8,1,4,2,32,0"
done
