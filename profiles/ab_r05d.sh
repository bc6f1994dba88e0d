#!/bin/bash
# Round 5: paired decoder launches (HDRSKY_DEC_PAIR, default 1) against the unpaired plan inside the step.
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-3}; STEPS=${2:-300}
run() { env $1 python bench.py --workload train --steps-only --steps $STEPS --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "HDRSKY_DEC_PAIR=1"
run "HDRSKY_DEC_PAIR=0"
done > $OUT/ab_d.txt 2>&1
cat $OUT/ab_d.txt
