#!/bin/bash
# Round 5, with the decoders paired (stream 0's backward chain ends ~350 us before stream 2's): where do the tail segments go?
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-2}; STEPS=${2:-300}
run() { env $1 python bench.py --workload train --steps-only --steps $STEPS --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-100s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "HDRSKY_X=default"
run "HDRSKY_PLAN_MOVE=wg_sunrad=1@wg_enc"
run "HDRSKY_PLAN_MOVE=wg_sunrad=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=bwd_sunrad=1@wg_enc,wg_sunrad=1@bwd_sunrad"
run "HDRSKY_PLAN_MOVE=apply_fc=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=apply_fc=1@wg_enc"
run "HDRSKY_PLAN_MOVE=wg_enc=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=wg_res=0@bwd_enc2,wg_enc=0@wg_res"
run "HDRSKY_PLAN_MOVE=wg_dec=0@bwd_enc2,wg_res=0@wg_dec,wg_enc=0@wg_res,wg_sunrad=1@disc_step"
run "HDRSKY_PLAN_MOVE=bwd_sunrad=1@disc_step,wg_sunrad=1@bwd_sunrad"
run "HDRSKY_BWD_DENSE_STREAM=0"
run "HDRSKY_VGG_SPLIT=0 HDRSKY_DISC_SPLIT=1"
done > $OUT/ab_e.txt 2>&1
cat $OUT/ab_e.txt
