#!/bin/bash
# A/B: consecutive segments of one stream merged into one hipGraph (HDRSKY_PLAN_MERGE): a graph launch costs its stream 13-19 us
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
M1="bwd_dense+bwd_sunpose+bwd_sunrad+wg_sunrad"; M2="wg_dec+wg_res"; M3="bwd_dec+bwd_res"; M4="bwd_enc+bwd_enc2"; M5="fwd_enc+zero"
for rep in 1 2; do
  echo "default: $(run)"
  echo "M1: $(HDRSKY_EXPERIMENTS=1 HDRSKY_PLAN_MERGE=$M1 run)"
  echo "M1,M2: $(HDRSKY_EXPERIMENTS=1 HDRSKY_PLAN_MERGE=$M1,$M2 run)"
  echo "M1,M2,M3: $(HDRSKY_EXPERIMENTS=1 HDRSKY_PLAN_MERGE=$M1,$M2,$M3 run)"
  echo "M1,M2,M3,M4: $(HDRSKY_EXPERIMENTS=1 HDRSKY_PLAN_MERGE=$M1,$M2,$M3,$M4 run)"
  echo "M1..M5: $(HDRSKY_EXPERIMENTS=1 HDRSKY_PLAN_MERGE=$M1,$M2,$M3,$M4,$M5 run)"
  echo "M4 only: $(HDRSKY_EXPERIMENTS=1 HDRSKY_PLAN_MERGE=$M4 run)"
done
