#!/bin/bash
# Round 5: the 64 px x 64 ch per WAVE tile (MI = NI = 4) on the 64->64 full-resolution class (VGG16 conv1_2 and its data gradient):
# +26 % saturated, -12 % alone (profiles/r05_conv_throughput_ni4.txt) - inside the step?
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-3}; STEPS=${2:-300}
run() { env $1 python bench.py --workload train --steps-only --steps $STEPS --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "HDRSKY_X=default"
run "HDRSKY_TILE_C64=4,1,4,4,32,1"
run "HDRSKY_TILE_C64=2,2,4,2,32,1"
done > $OUT/ab_g.txt 2>&1
cat $OUT/ab_g.txt
