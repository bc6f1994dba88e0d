#!/bin/bash
# Instruction-mix counters of the roofline kernel (res-block conv), rocprofv3 --pmc in two passes with --kernel-trace only.
# usage (on the GPU box, from the repo root): bash profiles/pmc_collect.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail > $R/gpurun_out/pmc_avail.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $R/gpurun_out/pmcA --output-format csv -- python3 $R/profiles/pmc_resconv.py > $R/gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmcB --output-format csv -- python3 $R/profiles/pmc_resconv.py > $R/gpurun_out/pmcB.log 2>&1
find $R/gpurun_out/pmcA $R/gpurun_out/pmcB -name "*counter_collection.csv" | head
