#!/bin/bash
# HBM-side bytes of ONE training step: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) over
# profiles/run_steps.py (six eager updating steps of bench.py's train workload), summed over every kernel and divided by six.
# usage (GPU box, repo root): bash profiles/pmc_step_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcs_$c --output-format csv -- python3 $R/profiles/run_steps.py 6 > $R/gpurun_out/pmcs_$c.json 2> $R/gpurun_out/pmcs_$c.log
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/pmcs_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    tot = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    res[c] = (tot, n)
steps = 6
tf, tw = sum(res["FETCH_SIZE"][0].values()), sum(res["WRITE_SIZE"][0].values())
print("per step: FETCH_SIZE %.1f MB (x2 for 16-byte streaming reads: <= %.1f MB), WRITE_SIZE %.1f MB" % (tf / steps / 1e3, 2 * tf / steps / 1e3, tw / steps / 1e3))
for k, v in sorted(res["FETCH_SIZE"][0].items(), key=lambda kv: -kv[1])[:24]:
    print("  %-28s fetch %8.1f MB  write %8.1f MB  launches/step %5.1f" % (k, v / steps / 1e3, res["WRITE_SIZE"][0].get(k, 0) / steps / 1e3, res["FETCH_SIZE"][1][k] / steps))
PY
find $R/gpurun_out -name "*counter_collection.csv" -delete; find $R/gpurun_out -name "*kernel_trace.csv" -path "*pmcs_*" -delete
