#!/bin/bash
# SQ counters of representative conv_igemm launches (profiles/pmc_conv.py), three passes of <= 8 SQ counters, kernel-trace only.
# usage (GPU box, repo root): bash profiles/pmc_conv.sh > gpurun_out/pmc_conv.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM"
P4="GRBM_GUI_ACTIVE GRBM_COUNT"
i=0
for p in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $p -d $R/gpurun_out/pmcconv_$i --output-format csv -- python3 $R/profiles/pmc_conv.py > $R/gpurun_out/pmcconv_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmcconv_$i.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for i in (1, 2, 3, 4):
    fs = glob.glob("$R/gpurun_out/pmcconv_%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if "conv_igemm_kernel" not in n:
            continue
        key = (n.split("conv_igemm_kernel")[1].split(">")[0] + ">", r.get("Grid_Size", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")), r.get("VGPR_Count", ""))
        agg.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in agg.items():
    print("conv_igemm%s grid %s lds %s vgpr %s" % key)
    for c, v in cs.items():
        v = sorted(v)
        print("    %-34s median %16.0f  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
find $R/gpurun_out -path "*pmcconv_*" -name "*.csv" -delete; find $R/gpurun_out -path "*pmcconv_*" -name "*.db" -delete
