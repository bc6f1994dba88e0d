#!/bin/bash
# A/B: extra dependencies between the plan's segments (HDRSKY_PLAN_DEPS): serialising phases that only contend
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default (vgg_target behind fwd_enc): $(run)"
  echo "VGG_TARGET_LATE=0: $(HDRSKY_EXPERIMENTS=1 HDRSKY_VGG_TARGET_LATE=0 run)"
  for d in disc_step:bwd_head disc_step:bwd_dense disc_step:bwd_dec loss_vgg_b:loss_vgg loss_vgg:loss_adv bwd_dense:loss_vgg_b wg_dec:bwd_res bwd_sunrad:disc_step apply_fc:bwd_sunpose; do
    echo "PLAN_DEPS=$d: $(HDRSKY_EXPERIMENTS=1 HDRSKY_PLAN_DEPS=$d run)"
  done
done
