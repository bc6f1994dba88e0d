#!/bin/bash
# HBM-side traffic of the round-3 kernels: rocprofv3 --pmc, one counter per pass, with --kernel-trace only.
# usage (GPU box, repo root): bash profiles/pmc_collect_r03.sh ; outputs under gpurun_out/pmc3_* (summarised by hand into profiles/r03_pmc_kernels.json)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc3_$c --output-format csv -- python3 $R/profiles/pmc_wgrad3.py > $R/gpurun_out/pmc3_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/pmc3_%s/**/*counter_collection.csv" % c, recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        if any(k in n for k in ("conv_wgrad", "wgrad_reduce", "dog_fused")):
            agg[(n, r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for (n, g), v in agg.items():
        v = sorted(v)
        out.setdefault(n + " grid " + g, {})[c + "_KB_median"] = v[len(v) // 2]
        out[n + " grid " + g]["launches"] = len(v)
print(json.dumps(out, indent=1))
PY
