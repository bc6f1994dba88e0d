#!/bin/bash
# A/B on the distortion-aware 128x512 step (res blocks + decoders): workgroup target of conv_wgrad2_kernel (its 1x1 form on the gathered operand)
run() { python3 bench.py --workload hires-train --da res,decoders --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default (192): $(run)"
  for v in 128 256 384 512; do echo "WGRAD2_WGS=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_WGRAD2_WGS=$v run)"; done
  echo "VGG_TARGET_LATE=0: $(HDRSKY_EXPERIMENTS=1 HDRSKY_VGG_TARGET_LATE=0 run)"
  echo "FC/NAB n/a; DA_MAT=0: $(HDRSKY_DA_MAT=0 run)"
done
