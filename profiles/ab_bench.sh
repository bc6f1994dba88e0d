#!/bin/bash
# A/B of bench.py variants back to back on ONE box (different boxes differ by +-2 %): usage  bash profiles/ab_bench.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --workload train --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train %.4f ms' % (d['ms_per_step'],))"; }
for rep in 1 2 3; do
run "Dense update on stream 1 (before)" HDRSKY_APPLY_FC_STREAM=1
run "on stream 0, behind bwd_enc" HDRSKY_APPLY_FC_STREAM=0
run "on stream 2, behind wg_sunrad" HDRSKY_APPLY_FC_STREAM=2
done
