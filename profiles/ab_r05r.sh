#!/bin/bash
# A/B of candidate tile-table entries on the round's final conv kernel (HDRSKY_TILE_RULES; profiles/r05_tile_sweep_v2.txt)
run() { python3 bench.py --steps 80 --warmup 8 --no-cpu-baseline --no-roofline-top --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd']['ms_per_step'])"; }
R1="256,512,4096,4096,256,512,3,-1=2,4,2,1,32,1"                      # vgg conv3_2/3 (+ data gradients) on a half batch: 64 px x 64 ch
R2="64,127,65536,131072,1,8,3,-1=2,2,4,2,32,1"                        # vgg conv1_1 (3->64)
R3="64,127,131072,131072,64,64,3,0=4,1,4,4,32,1"                      # vgg conv1_2 on the whole batch
R5="64,127,65536,262143,1,8,4,-1=2,2,4,2,32,1"                        # dis.d1 (6->64, 4x4 stride 2)
R6="17,63,32768,32768,64,64,3,0=4,2,4,1,32,1"                         # 64->32 data gradient at 16x64
for rep in 1 2; do
  echo "default: $(run)"
  echo "R1: $(HDRSKY_EXPERIMENTS=1 HDRSKY_TILE_RULES="$R1" run)"
  echo "R2: $(HDRSKY_EXPERIMENTS=1 HDRSKY_TILE_RULES="$R2" run)"
  echo "R3: $(HDRSKY_EXPERIMENTS=1 HDRSKY_TILE_RULES="$R3" run)"
  echo "R5: $(HDRSKY_EXPERIMENTS=1 HDRSKY_TILE_RULES="$R5" run)"
  echo "R6: $(HDRSKY_EXPERIMENTS=1 HDRSKY_TILE_RULES="$R6" run)"
  echo "all: $(HDRSKY_EXPERIMENTS=1 HDRSKY_TILE_RULES="$R1;$R2;$R3;$R5;$R6" run)"
done
