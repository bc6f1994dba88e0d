#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --steps 40 --workload train --da $DA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms' % (d['ms_per_step'],))"; }
for rep in 1 2; do
DA=res run "da res: global gather" HDRSKY_DA_REGION=0
DA=res run "da res: region" X=1
DA=all run "da all: global gather" HDRSKY_DA_REGION=0
DA=all run "da all: region" X=1
done
