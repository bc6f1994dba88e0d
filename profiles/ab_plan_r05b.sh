#!/bin/bash
# Round 5: the perceptual term's prediction pass as ONE batch-32 pass (fewer, fatter launches: 250 us alone against 2 x 220 us for
# the two half batches) with the stream it frees used for the real half of the discriminator step.
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-2}; STEPS=${2:-300}
run() { env $1 python bench.py --workload train --steps-only --steps $STEPS --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-90s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "HDRSKY_X=default"
run "HDRSKY_VGG_SPLIT=0"
run "HDRSKY_VGG_SPLIT=0 HDRSKY_DISC_SPLIT=1"
run "HDRSKY_DISC_SPLIT=1"
run "HDRSKY_VGG_SPLIT=0 HDRSKY_PLAN_MOVE=loss_adv=2@vgg_target"
run "HDRSKY_NAB_ONE=0"
done > $OUT/ab_b.txt 2>&1
cat $OUT/ab_b.txt
