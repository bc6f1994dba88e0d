#!/bin/bash
# A/B: the plan-level switches once more, after the footprint re-tuning shifted the balance of the three streams
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default: $(run)"
  for kv in APPLY_AFTER_FC=1 WG_ENC_SPLIT=1 WG_RES_STREAM=2 WG_RES_STREAM=0 APPLY_FC_STREAM=1 BWD_DENSE_STREAM=1 VGG_SPLIT=0 DISC_SPLIT=1 SUN3=1 DEC_HEAD_EARLY=0 INXF_AFFINE_MIN=16 INXF_AFFINE_MIN=256 "PLAN_MOVE=wg_sunrad=1@wg_res" "PLAN_MOVE=apply=1@wg_res"; do
    echo "$kv: $(env HDRSKY_EXPERIMENTS=1 "HDRSKY_$kv" bash -c "$(declare -f run); run")"
  done
done
