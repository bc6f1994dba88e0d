#!/bin/bash
# A/B: workgroup targets of the weight-gradient kernels inside the step (fewer pixel chunks = smaller partial slabs and reduce)
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default: $(run)"
  for v in 128 192 320 384 512; do echo "WGRAD2_WGS=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_WGRAD2_WGS=$v run)"; done
  for v in 128 384 512; do echo "WGRAD3_WGS=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_WGRAD3_WGS=$v run)"; done
done
