#!/bin/bash
export HDRSKY_EXPERIMENTS=1   # the tuning hooks this script sets are behind the gate since round 4 (csrc/hooks.h, hooks.py)
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --workload train --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms' % (d['ms_per_step'],))"; env "$@" python bench.py --no-cpu-baseline --workload fwd --steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fwd   %.4f ms' % (d['ms_per_step'],))"; }
for rep in 1 2; do
run "4x16-pixel layers with Cout >= 128: 32 px x 128 ch, 8 waves (shipping)" A=1
run "32 px x 64 ch, 4 waves" HDRSKY_TILE_T16=1,4,2,1,16,1
run "64 px x 64 ch, 4 waves" HDRSKY_TILE_T16=1,4,4,1,16,1
done
