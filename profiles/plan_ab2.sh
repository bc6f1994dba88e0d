# A/B of the Dense update's placement and width (trainer.py: apply_fc; csrc/fc_update.hip: HDRSKY_FC_UPDATE_WGS), same box
run() { env $1 python bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-90s' % '$1', d['ms_per_step'])"; }
for rep in 1 2; do
run "HDRSKY_X=default"
run "HDRSKY_FC_UPDATE_WGS=256"
run "HDRSKY_FC_UPDATE_WGS=512"
run "HDRSKY_FC_UPDATE_WGS=64 HDRSKY_APPLY_FC_STREAM=3 HDRSKY_PLAN_MOVE=apply_fc=3@bwd_dense"
run "HDRSKY_FC_UPDATE_WGS=128 HDRSKY_APPLY_FC_STREAM=3 HDRSKY_PLAN_MOVE=apply_fc=3@bwd_dense"
run "HDRSKY_FC_UPDATE_WGS=256 HDRSKY_APPLY_FC_STREAM=3 HDRSKY_PLAN_MOVE=apply_fc=3@bwd_dense"
done
