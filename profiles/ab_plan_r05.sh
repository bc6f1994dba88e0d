#!/bin/bash
# Round 5: where the tail segments of the step run once the main chain (stream 0) got shorter (one-launch InstanceNorm backward,
# grouped decoder launches): stream 0 finishes its backward chain ~150-300 us before streams 1 / 2 finish theirs.
# HDRSKY_PLAN_MOVE="segment=stream@after": the segment goes to `stream`, enqueued right behind segment `after` (dependencies kept).
# usage (GPU box): bash profiles/ab_plan_r05.sh [reps] [steps]
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-2}; STEPS=${2:-300}
run() { env $1 python bench.py --workload train --steps-only --steps $STEPS --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-90s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "HDRSKY_X=default"
run "HDRSKY_PLAN_MOVE=wg_sunrad=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=wg_enc=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=wg_sunrad=0@bwd_enc2,wg_enc=0@wg_sunrad"
run "HDRSKY_PLAN_MOVE=wg_res=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=apply_fc=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=apply_fc=2@bwd_dense"
run "HDRSKY_PLAN_MOVE=apply_fc=2@bwd_sunpose"
run "HDRSKY_PLAN_MOVE=wg_sunrad=1@wg_enc"
run "HDRSKY_PLAN_MOVE=bwd_sunrad=0@bwd_enc2,wg_sunrad=0@bwd_sunrad"
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
