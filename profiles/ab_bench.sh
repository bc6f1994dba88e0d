#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --workload train --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms' % (d['ms_per_step']))"; }
for rep in 1 2; do
run "three streams" X=1
run "four streams (weight-gradient segments on their own)" HDRSKY_STREAMS=4
done
