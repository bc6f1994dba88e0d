#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --workload fwd --da all --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fwd --da all %.4f ms' % d['ms_per_step'])"; }
for rep in 1 2; do
run "default" X=1
run "one tap per round" HDRSKY_DA_TPR=1
run "two taps per round" HDRSKY_DA_TPR=2
done
