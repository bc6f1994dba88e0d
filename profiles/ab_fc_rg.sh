#!/bin/bash
# Round 4: reduction groups per workgroup of fc_mfma_kernel (HDRSKY_FC_RG, tuning hook) and reduction slices (HDRSKY_FC_NSPLIT):
# the fc1 microbench, then the forward pass and the training step.   usage (GPU box): bash profiles/ab_fc_rg.sh
export HDRSKY_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_fc; mkdir -p $OUT
{
for rg in 1 2 4; do for ns in 4 8; do
  echo "== HDRSKY_FC_RG=$rg HDRSKY_FC_NSPLIT=$ns"
  HDRSKY_FC_RG=$rg HDRSKY_FC_NSPLIT=$ns python profiles/microbench_fc.py 2>/dev/null
done; done
run() { env $1 python bench.py --workload all --no-cpu-baseline --no-parity --no-roofline-top --steps 200 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s train %.4f ms  fwd %.4f ms' % ('$1', d.get('ms_per_step'), d.get('fwd',{}).get('ms_per_step')))"; }
for rep in 1 2; do
run "HDRSKY_FC_RG=1"
run "HDRSKY_FC_RG=2"
run "HDRSKY_FC_RG=4"
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
