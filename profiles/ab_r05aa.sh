#!/bin/bash
# A/B: the Dense update (HBM-bound, 1 GB) moved from the end of the step to a fourth stream confined to a slice of the chip, where it can
# start as soon as bwd_dense is done (HDRSKY_APPLY_FC_CUS="lo,hi[,step]")
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default: $(run)"
  for m in "0,32" "0,64" "0,128" "0,256,8" "0,256,4" "0,256,2" "0,256"; do
    echo "APPLY_FC_CUS=$m: $(HDRSKY_EXPERIMENTS=1 HDRSKY_APPLY_FC_CUS=$m run)"
  done
done
