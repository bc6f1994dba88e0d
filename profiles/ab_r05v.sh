#!/bin/bash
# A/B: the other grid-size knobs re-tuned INSIDE the step (they were chosen on launches timed alone)
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "default: $(run)"
  for kv in NAB_TARGET=256 NAB_TARGET=384 NAB_TARGET=768 FC_NSPLIT=2 FC_NSPLIT=8 FC_RG=2 FC_UPDATE_NB=1 WGRAD2_MINT=1 WGRAD2_MINT=4 WGRAD3_MINPX=128 WGRAD3_MINPX=512; do
    echo "$kv: $(env HDRSKY_EXPERIMENTS=1 HDRSKY_$kv bash -c "$(declare -f run); run")"
  done
done
