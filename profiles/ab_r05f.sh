#!/bin/bash
# Round 5: the Dense kernels' update deferred into the next replay's forward pass (bench.py default) against closing its own step.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-3}; STEPS=${2:-300}
run() { python bench.py --workload train --steps-only --steps $STEPS --warmup 20 $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "--defer-dense"
run ""
done > $OUT/ab_f.txt 2>&1
cat $OUT/ab_f.txt
