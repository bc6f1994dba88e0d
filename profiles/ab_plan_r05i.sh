#!/bin/bash
# Round 5, final kernels (paired decoders, nt Dense images): stream 0 idles ~300 us before grads_ready (profiles/r05_segment_timeline.txt).
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-2}; STEPS=${2:-300}
run() { env $1 python bench.py --workload train --steps-only --steps $STEPS --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-110s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "HDRSKY_X=default"
run "HDRSKY_PLAN_MOVE=wg_enc=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=wg_dec=0@bwd_dec"
run "HDRSKY_PLAN_MOVE=wg_res=0@bwd_res"
run "HDRSKY_PLAN_MOVE=wg_res=0@bwd_res,wg_enc=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=wg_dec=0@bwd_dec,wg_res=0@bwd_res,wg_enc=0@bwd_enc2"
run "HDRSKY_PLAN_MOVE=wg_res=0@bwd_res,wg_enc=0@bwd_enc2,wg_sunrad=1@wg_dec"
run "HDRSKY_PLAN_MOVE=wg_res=0@bwd_res,wg_enc=0@bwd_enc2,bwd_sunrad=1@wg_dec,wg_sunrad=1@bwd_sunrad"
run "HDRSKY_PLAN_MOVE=wg_enc=0@bwd_enc2,wg_sunrad=1@wg_res"
run "HDRSKY_PLAN_MOVE=wg_res=0@bwd_enc2,wg_enc=0@wg_res,apply_fc=2@bwd_sunpose"
done > $OUT/ab_i.txt 2>&1
cat $OUT/ab_i.txt
