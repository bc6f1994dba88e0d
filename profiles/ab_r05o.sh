#!/bin/bash
# A/B: LDS of the conv epilogue tile (HDRSKY_CONV_EPI_LDS KB: own LDS while planes + tile fit / aliased onto the operand planes)
run() { python3 bench.py --workload $1 --steps 60 --warmup 5 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('fwd',{}).get('ms_per_step'))"; }
for rep in 1 2; do
  for v in 80 48 40 0 160; do
    echo "train HDRSKY_CONV_EPI_LDS=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_CONV_EPI_LDS=$v run all)"
  done
done
for v in 80 40 0; do echo "hires-train HDRSKY_CONV_EPI_LDS=$v: $(HDRSKY_EXPERIMENTS=1 HDRSKY_CONV_EPI_LDS=$v run hires-train)"; done
