#!/bin/bash
# A/B: the Dense update as a TRICKLE - a capped grid whose workgroups walk their k tiles (HDRSKY_FC_UPDATE_ROWS x 32 column blocks) - on a
# fourth stream behind bwd_dense (HDRSKY_APPLY_FC_CUS=-1), i.e. beside 0.8 ms of backward pass instead of at the end of the step
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default: $(run)"
  for r in 1 2 4 8 16 32 64; do
    echo "early, ROWS=$r: $(HDRSKY_EXPERIMENTS=1 HDRSKY_APPLY_FC_CUS=-1 HDRSKY_FC_UPDATE_ROWS=$r run)"
  done
  for r in 8 32; do echo "at the end, ROWS=$r: $(HDRSKY_EXPERIMENTS=1 HDRSKY_FC_UPDATE_ROWS=$r run)"; done
done
