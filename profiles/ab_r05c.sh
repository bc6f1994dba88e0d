#!/bin/bash
# Round 5: the saturated-regime tile table (HDRSKY_TILE_TABLE=5, default) against round 4's (=4), and the one-launch InstanceNorm
# backward against the sliced form, inside the step.   usage (GPU box): bash profiles/ab_r05c.sh [reps] [steps]
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-3}; STEPS=${2:-300}
run() { env $1 python bench.py --workload ${WL:-train} --steps-only --steps $STEPS --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %s %.4f ms' % ('$1', '${WL:-train}', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "HDRSKY_TILE_TABLE=5"
run "HDRSKY_TILE_TABLE=4"
run "HDRSKY_TILE_TABLE=5 HDRSKY_NAB_ONE=0"
WL=fwd run "HDRSKY_TILE_TABLE=5"
WL=fwd run "HDRSKY_TILE_TABLE=4"
done > $OUT/ab_c.txt 2>&1
cat $OUT/ab_c.txt
