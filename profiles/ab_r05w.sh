#!/bin/bash
# A/B: fused Dense update with one 32x32 block per wave (58 VGPRs, 8 waves per SIMD, twice the workgroups) against two, inside the step
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "default (NB=2, nt=1): $(run)"
  echo "NB=1 nt=1: $(HDRSKY_EXPERIMENTS=1 HDRSKY_FC_UPDATE_NB=1 run)"
  echo "NB=1 nt=0: $(HDRSKY_EXPERIMENTS=1 HDRSKY_FC_UPDATE_NB=1 HDRSKY_FC_NT=0 run)"
done
for nb in 2 1; do echo "alone, NB=$nb:"; HDRSKY_FC_UPDATE_NB=$nb python3 profiles/microbench_fc_nt.py 2>&1 | grep "FC_NT=[01] " | tail -2; done
