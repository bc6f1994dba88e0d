#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms  fwd %.4f ms' % (d['ms_per_step'], d['fwd']['ms_per_step']))"; }
for rep in 1 2; do
run "no priorities" HDRSKY_SIDE_PRIORITY=0
run "side stream high priority" X=1
run "train: stream 0 high" HDRSKY_STREAM_PRIORITY=-1,0,0
run "train: stream 1 high" HDRSKY_STREAM_PRIORITY=0,-1,0
run "train: stream 0,1 high" HDRSKY_STREAM_PRIORITY=-1,-1,0
done
