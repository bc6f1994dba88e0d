#!/usr/bin/env python3
"""Per-queue kernel timeline of ONE generator forward pass from a rocprofv3 --kernel-trace CSV of `bench.py --workload fwd`
(a pass ends with blend_kernel): fwd_timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# find forward boundaries: blend_kernel ends a forward
ends = [i for i, r in enumerate(rows) if "blend_kernel" in r["Kernel_Name"]]
k = len(ends) // 2
a, b = ends[k - 1] + 1, ends[k] + 1
fw = rows[a:b]
t0 = fw[0]["s"]
print("forward: %d kernels, %.1f us" % (len(fw), (fw[-1]["e"] - t0) / 1e3))
qs = {}
for r in fw: qs.setdefault(r["Queue_Id"], []).append(r)
for q, v in qs.items():
    print("== queue", q, "kernels", len(v), "busy %.1f" % (sum(r["e"] - r["s"] for r in v) / 1e3), "last end %.1f" % ((v[-1]["e"] - t0) / 1e3))
    prev = None
    for r in v:
        gap = (r["s"] - prev) / 1e3 if prev else 0
        print("  %7.1f %6.1f gap %5.1f %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, gap, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]))
        prev = r["e"]
