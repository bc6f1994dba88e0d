# A/B of the optimizer segments' placement (trainer.py: apply_fc, apply), same box, three rounds
run() { env $1 python bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-80s' % '$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
run "HDRSKY_X=default"
run "HDRSKY_APPLY_FC_STREAM=3"
run "HDRSKY_APPLY_FC_STREAM=3 HDRSKY_PLAN_MOVE=apply_fc=3@bwd_dense"
run "HDRSKY_APPLY_FC_STREAM=1 HDRSKY_APPLY_AFTER_FC=1"
done
