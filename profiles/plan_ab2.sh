# A/B of the optimizer segments' placement (trainer.py: apply_fc / apply), same box, two rounds
run() { env $1 python bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s' % '$1', d['ms_per_step'])"; }
for rep in 1 2; do
run "HDRSKY_X=default"
run "HDRSKY_APPLY_FC_STREAM=1 HDRSKY_APPLY_AFTER_FC=1"
run "HDRSKY_APPLY_FC_STREAM=1"
run "HDRSKY_APPLY_AFTER_FC=1"
done
