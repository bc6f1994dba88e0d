#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms  fwd %.4f ms' % (d['ms_per_step'], d['fwd']['ms_per_step']))"; }
for rep in 1 2; do
run "default 2,4,4,1,32,1 (128 px x 64 ch)" X=1
run "2,4,2,1,32,1 (64 px x 64 ch)" HDRSKY_TILE_A=2,4,2,1,32,1
run "1,8,4,1,32,1 (64 px x 128 ch)" HDRSKY_TILE_A=1,8,4,1,32,1
run "2,4,4,2,32,1 (128 px x 128 ch)" HDRSKY_TILE_A=2,4,4,2,32,1
done
