#!/bin/bash
# A/B of further tile entries for the 128x512 network (mixed in the sweep: slower alone, faster saturated), injected with HDRSKY_TILE_RULES
run() { python3 bench.py --workload hires-train --steps 40 --warmup 5 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
R1="256,512,8192,8192,128,512,-1,-1=2,4,4,2,32,1"
R2="128,255,16384,16384,256,256,3,-1=2,4,4,2,32,1"
R3="128,255,32768,32768,64,64,3,-1=2,2,4,4,32,1"
R4="128,255,32768,32768,64,64,4,-1=1,8,4,1,32,1"
R5="64,127,65536,65536,128,128,4,-1=4,1,4,4,32,1"
R6="64,127,131072,131072,32,32,3,0=4,1,4,4,32,1"
for rep in 1 2; do
  echo "default: $(run)"
  for n in 1 2 3 4 5 6; do eval r=\$R$n; echo "R$n: $(HDRSKY_EXPERIMENTS=1 HDRSKY_TILE_RULES="$r" run)"; done
  echo "all: $(HDRSKY_EXPERIMENTS=1 HDRSKY_TILE_RULES="$R1;$R2;$R3;$R4;$R5;$R6" run)"
done
