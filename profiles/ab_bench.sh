#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms  fwd %.4f ms' % (d['ms_per_step'], d['fwd']['ms_per_step']))"; }
for rep in 1 2 3; do
run "sun decoder's deconvolutions in fwd_blend (before)" HDRSKY_DEC_HEAD_EARLY=0
run "in fwd_enc" X=1
done
