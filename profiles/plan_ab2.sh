# A/B: where the sun-radiance head's weight gradients run (stream 2's chain ends the step), same box, three rounds
run() { env $1 python bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s' % '$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
run "HDRSKY_X=default"
run "HDRSKY_PLAN_MOVE=wg_sunrad=1@wg_res"
run "HDRSKY_PLAN_MOVE=wg_sunrad=1@wg_enc"
run "HDRSKY_PLAN_MOVE=wg_sunrad=1@wg_dec"
done
