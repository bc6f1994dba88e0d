#!/bin/bash
# A/B of the round's default changes on the 128x512 workloads (forward + losses, training step): each line = ms per step.
run() { python3 bench.py --workload $1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for w in hires hires-train; do
  for rep in 1 2; do
    echo "$w default: $(run $w)"
    echo "$w HDRSKY_TILE_TABLE=4: $(HDRSKY_TILE_TABLE=4 run $w)"
    echo "$w HDRSKY_DEC_PAIR=0: $(HDRSKY_DEC_PAIR=0 run $w)"
    echo "$w FC_W_NT=0 FC_NT=0: $(HDRSKY_EXPERIMENTS=1 HDRSKY_FC_W_NT=0 HDRSKY_FC_NT=0 run $w)"
    echo "$w NAB_ONE=0: $(HDRSKY_NAB_ONE=0 run $w)"
  done
done
