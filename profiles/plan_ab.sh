export HDRSKY_EXPERIMENTS=1   # the tuning hooks this script sets are behind the gate since round 4 (csrc/hooks.h, hooks.py)
run() { HDRSKY_PLAN_MOVE="$1" python bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s' % '$1', d['ms_per_step'])"; }
for rep in 1 2; do
run ""
run "wg_res=0@bwd_enc"
run "wg_dec=0@bwd_enc"
run "wg_dec=0@bwd_enc,wg_res=0@wg_dec"
run "wg_sunrad=0@bwd_enc"
run "wg_res=0@bwd_enc,wg_sunrad=1@wg_dec"
run "wg_res=2@wg_sunrad"
run "bwd_sunrad=1@disc_step,wg_sunrad=1@bwd_sunrad,wg_res=0@bwd_enc"
done
