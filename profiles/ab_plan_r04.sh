#!/bin/bash
# NOTE (round 5): HDRSKY_FC_UPDATE_WGS named a persistent-workgroup throttle of fc_xtdy_kernel that was removed before round 4's
# final commit (csrc/hooks.h does not read it): the rows of profiles/r04_plan_ab.txt that carry it repeat their neighbours
# without it and are NOT reproducible with the committed code.  The HDRSKY_PLAN_MOVE rows are.
# Round 4: the Dense update (apply_fc: ~320 us, HBM-bound, the tail of the step on stream 2) on a stream of its own right
# behind bwd_dense, as every tile's workgroup or throttled to a fraction of the chip (HDRSKY_FC_UPDATE_WGS persistent
# workgroups), so that it hides beside the backward pass.   usage (GPU box): bash profiles/ab_plan_r04.sh
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_plan; mkdir -p $OUT
run() { env $1 python bench.py --workload train --no-cpu-baseline --no-parity --no-roofline-top --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in 1 2; do
run "HDRSKY_X=default"
run "HDRSKY_PLAN_MOVE=apply_fc=3@wg_dense"
run "HDRSKY_PLAN_MOVE=apply_fc=3@wg_dense HDRSKY_FC_UPDATE_WGS=1024"
run "HDRSKY_PLAN_MOVE=apply_fc=3@wg_dense HDRSKY_FC_UPDATE_WGS=512"
run "HDRSKY_PLAN_MOVE=apply_fc=3@wg_dense HDRSKY_FC_UPDATE_WGS=256"
run "HDRSKY_PLAN_MOVE=apply_fc=3@wg_dense HDRSKY_FC_UPDATE_WGS=128"
run "HDRSKY_PLAN_MOVE=apply_fc=3@bwd_res HDRSKY_FC_UPDATE_WGS=512"
run "HDRSKY_FC_UPDATE_WGS=1024"
done > $OUT/ab2.txt 2>&1
cat $OUT/ab2.txt
