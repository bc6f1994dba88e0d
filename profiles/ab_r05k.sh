#!/bin/bash
# Round 5: non-temporal policies beyond the Dense images - the Dense layers' weight stream (fc_mfma_kernel loads, HDRSKY_FC_W_NT)
# and the conv-side optimizer's w / ms stores (HDRSKY_OPT_NT) - inside the step and in the forward pass.
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --workload ${WL:-train} --steps-only --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %s %.4f ms' % ('$1', '${WL:-train}', d.get('ms_per_step')))"; }
for rep in 1 2 3; do
run "HDRSKY_X=default"
run "HDRSKY_FC_W_NT=1"
run "HDRSKY_OPT_NT=1"
run "HDRSKY_FC_W_NT=1 HDRSKY_OPT_NT=1"
WL=fwd run "HDRSKY_X=default"
WL=fwd run "HDRSKY_FC_W_NT=1"
done
