#!/bin/bash
# A/B: Dense-layer workgroup shape (reduction groups per workgroup, reduction slices) on the forward pass
run() { python3 bench.py --workload fwd --steps 200 --warmup 20 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('fwd',d)['ms_per_step'])"; }
for rep in 1 2; do
  echo "default: $(run)"
  for kv in FC_RG=1 FC_RG=2 FC_NSPLIT=2 FC_NSPLIT=8 "FC_RG=2 HDRSKY_FC_NSPLIT=8" "FC_RG=1 HDRSKY_FC_NSPLIT=8" NAB_TARGET=256; do
    echo "$kv: $(env HDRSKY_EXPERIMENTS=1 HDRSKY_$kv bash -c "$(declare -f run); run")"
  done
done
