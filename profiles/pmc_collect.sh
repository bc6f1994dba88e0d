#!/bin/bash
# HBM-side traffic and MFMA counters of the roofline kernel (sample-resident res-block launch), rocprofv3 --pmc in separate
# passes with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (on the GPU box, from the repo root): bash profiles/pmc_collect.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$tag --output-format csv -- python3 $R/profiles/pmc_resconv.py > $R/gpurun_out/pmc_$tag.log 2>&1
done
find $R/gpurun_out -name "*counter_collection.csv" | head
