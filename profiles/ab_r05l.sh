export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --workload fwd --steps-only --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s fwd %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in 1 2 3; do
run "HDRSKY_X=default"
run "HDRSKY_FWD_STAGGER=1"
done
