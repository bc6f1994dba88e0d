#!/bin/bash
# Round 5: non-temporal stores of the Dense layers' bf16 images in the fused update (HDRSKY_FC_NT=1, default) against default-policy
# stores (=0), inside the step; and the roofline_hbm rows alone.
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_plan; mkdir -p $OUT
REPS=${1:-3}; STEPS=${2:-300}
run() { env $1 python bench.py --workload train --steps-only --steps $STEPS --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %.4f ms' % ('$1', d.get('ms_per_step')))"; }
for rep in $(seq $REPS); do
run "HDRSKY_FC_NT=1"
run "HDRSKY_FC_NT=0"
run "HDRSKY_FC_NT=5"
done > $OUT/ab_h.txt 2>&1
cat $OUT/ab_h.txt
python profiles/microbench_fc_update.py 2>&1 | grep -v amdgpu.ids | tee $OUT/microbench_fc_update.txt
