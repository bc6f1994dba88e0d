#!/bin/bash
# Round 4, VERDICT item 3(i): raw conv outputs in front of InstanceNorm / BatchNorm stored as bf16 (HDRSKY_RAW_BF16, tuning
# hook) against fp32 storage - bench.py's training step and forward pass back to back on ONE box, twice, and the per-kernel
# statistics of six eager steps under each setting.   usage:  bash profiles/ab_raw_bf16.sh  (on the GPU box)
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_raw16; mkdir -p $OUT
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu-baseline --no-parity --no-roofline-top --workload train --steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train %.4f ms' % (d['ms_per_step'],))"; }
for rep in 1 2; do
run "raw conv outputs as bf16" HDRSKY_RAW_BF16=1
run "raw conv outputs as fp32" HDRSKY_RAW_BF16=0
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
export TMPDIR=/tmp
for v in 1 0; do
  export HDRSKY_RAW_BF16=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$v -o p -- python3 profiles/run_steps.py 6 > $OUT/prof$v.log 2>&1
  f=$(find $OUT/prof$v -name '*kernel_stats.csv' | head -1)
  cp "$f" $OUT/kernel_stats_raw$v.csv
  rm -rf $OUT/prof$v
done
unset HDRSKY_RAW_BF16
python3 profiles/segment_timeline.py 2>&1 | grep -v amdgpu.ids > $OUT/segment_timeline.txt
