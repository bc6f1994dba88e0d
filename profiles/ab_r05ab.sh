#!/bin/bash
# as ab_r05aa.sh with GPU_MAX_HW_QUEUES=8 (the fourth trainer stream otherwise shares a hardware queue: 4.2 ms)
run() { python3 bench.py --workload train --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
export GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
  echo "Q8 default: $(run)"
  for m in "0,32" "0,64" "0,128" "0,256,4" "0,256,2" "0,256"; do
    echo "Q8 APPLY_FC_CUS=$m: $(HDRSKY_EXPERIMENTS=1 HDRSKY_APPLY_FC_CUS=$m run)"
  done
done
