#!/bin/bash
# A/B: the deferred Dense update (Trainer(defer_dense=True)) again, now that the update runs on small-footprint workgroups
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "default: $(run)"
  echo "--defer-dense: $(run --defer-dense)"
done
