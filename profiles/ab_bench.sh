#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --steps 40 --workload train --da $DA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms' % (d['ms_per_step'],))"; }
for rep in 1 2; do
DA=res run "da res: generic wgrad" HDRSKY_DA_WGRAD_REGION=0
DA=res run "da res: region wgrad all" HDRSKY_DA_WGRAD_REGION_MAXC=256
DA=res run "da res: 64-pixel tiles" HDRSKY_DA_TM=64
DA=all run "da all: generic wgrad" HDRSKY_DA_WGRAD_REGION=0
DA=all run "da all: region wgrad C<=64" X=1
DA=all run "da all: region wgrad C<=32" HDRSKY_DA_WGRAD_REGION_MAXC=32
DA=all run "da all: region wgrad all" HDRSKY_DA_WGRAD_REGION_MAXC=256
done
