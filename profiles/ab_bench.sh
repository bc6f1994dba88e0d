#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --workload train --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms' % (d['ms_per_step']))"; }
for rep in 1 2; do
run "alone 128 group 192" HDRSKY_WGRAD=128,0,192
run "alone 96 group 192" HDRSKY_WGRAD=96,0,192
run "alone 64 group 192" HDRSKY_WGRAD=64,0,192
run "alone 128 group 224" HDRSKY_WGRAD=128,0,224
run "alone 128 group 256" HDRSKY_WGRAD=128,0,256
run "alone 112 group 208" HDRSKY_WGRAD=112,0,208
done
