#!/bin/bash
# as ab_so3.sh, on the train workload only (builds that lack a path the forward pass needs)
R=$1; shift
P=$(ls -d *_amd)
for i in $(seq $R); do
  for v in "$@"; do
    cp ab/$v.so $P/libhdrsky.so
    python3 bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])" || exit 1
  done
done
cp ab/$1.so $P/libhdrsky.so
