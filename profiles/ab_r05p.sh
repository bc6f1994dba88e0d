#!/bin/bash
# A/B: conv_igemm touches its halo patch ahead of the prologue (HDRSKY_CONV_EARLY)
run() { python3 bench.py --workload $1 --steps 60 --warmup 5 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('fwd',{}).get('ms_per_step'))"; }
for l in "vgg1_2" "l2b"; do for v in 1 0; do echo "EARLY=$v bf16: $(HDRSKY_EXPERIMENTS=1 HDRSKY_CONV_EARLY=$v STAMP_BF16=1 python3 profiles/stamp_conv.py "$l" 2>&1 | grep -v amdgpu.ids)"; done; done
for l in "res 128" "conv2_d" "l1b"; do for v in 1 0; do echo "EARLY=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_CONV_EARLY=$v python3 profiles/stamp_conv.py "$l" 2>&1 | grep -v amdgpu.ids)"; done; done
for rep in 1 2 3; do
  for v in 1 0; do
    echo "train HDRSKY_CONV_EARLY=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_CONV_EARLY=$v run all)"
  done
done
for v in 1 0; do echo "hires-train HDRSKY_CONV_EARLY=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_CONV_EARLY=$v run hires-train)"; done
