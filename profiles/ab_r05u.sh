#!/bin/bash
# A/B, second pass: workgroup target of conv_wgrad2_kernel (default 256) inside the step, 32x128 and 128x512
run() { python3 bench.py --workload $1 --steps $2 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "train default: $(run train 80)"
  for v in 160 176 192 208 224; do echo "train WGRAD2_WGS=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_WGRAD2_WGS=$v run train 80)"; done
  echo "train WGRAD2_WGS=192 WGRAD3_WGS=192: $(HDRSKY_EXPERIMENTS=1 HDRSKY_WGRAD2_WGS=192 HDRSKY_WGRAD3_WGS=192 run train 80)"
done
for rep in 1 2; do
  echo "hires-train default: $(run hires-train 40)"
  for v in 128 192 384 512; do echo "hires-train WGRAD2_WGS=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_WGRAD2_WGS=$v run hires-train 40)"; done
done
