#!/bin/bash
# Same-box A/B of several builds of libhdrsky.so (ab/NAME.so; `ab/` is scratch, *.so is git-ignored): alternates them R times on
# bench.py's default workload and prints ms_per_step / forward ms of every run.
# usage (GPU box, repo root): bash profiles/ab_so3.sh R NAME NAME ...   (the first NAME is restored at the end)
R=$1; shift
P=$(ls -d *_amd)
for i in $(seq $R); do
  for v in "$@"; do
    cp ab/$v.so $P/libhdrsky.so
    python3 bench.py --no-cpu-baseline --no-roofline-top --no-parity --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['fwd']['ms_per_step'])" || exit 1
  done
done
cp ab/$1.so $P/libhdrsky.so
