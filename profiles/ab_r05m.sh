#!/bin/bash
# Round 5: non-temporal operand copies in conv_wgrad2_kernel (HDRSKY_WGRAD2_NT: 1 x, 2 dy) and non-temporal x loads in the one-launch
# InstanceNorm backward (HDRSKY_NAB_NT), inside the step.
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --workload train --steps-only --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in 1 2 3; do
run "HDRSKY_X=default"
run "HDRSKY_WGRAD2_NT=1"
run "HDRSKY_WGRAD2_NT=3"
run "HDRSKY_NAB_NT=1"
done
