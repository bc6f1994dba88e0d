#!/bin/bash
# Same-box A/B of two builds of libhdrsky.so (ab/base.so, ab/new.so; `ab/` is scratch, *.so is git-ignored): alternates the
# two libraries R times on bench.py's train workload and prints ms_per_step of every run.
# usage (GPU box, repo root): bash profiles/ab_so.sh [R=3] [extra bench args]
R=${1:-3}; shift
P=$(ls -d *_amd)
for i in $(seq $R); do
  for v in base new; do
    cp ab/$v.so $P/libhdrsky.so
    python3 bench.py --workload train --no-cpu-baseline --no-roofline-top --no-parity --steps 300 --warmup 30 "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])" || exit 1
  done
done
cp ab/new.so $P/libhdrsky.so
