#!/bin/bash
# A/B: priorities of the training step's three streams (HDRSKY_STREAM_PRIO, torch: -1 = high)
run() { python3 bench.py --workload train --steps 60 --warmup 5 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  for v in "0,0,0" "-1,0,0" "0,-1,-1" "-1,-1,0" "0,0,-1" "0,-1,0"; do
    echo "HDRSKY_STREAM_PRIO=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_STREAM_PRIO=$v run)"
  done
done
