#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms  fwd %.4f ms  roof %.3f us' % (d['ms_per_step'], d['fwd']['ms_per_step'], d['roofline']['avg_launch_us']))"; }
for rep in 1 2; do
run "resize-deconvolutions on materialised bf16 operands" X=1
run "resize fused into the conv staging" HDRSKY_DECONV_MAT=0
done
