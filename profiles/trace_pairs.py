import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    short = "v3" if "conv_wgrad3" in n else ("v2" if "conv_wgrad2" in n else ("v1" if "conv_wgrad_kernel" in n else ("reduce" if "wgrad_reduce" in n else None)))
    if short is None:
        prev = None; continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if short == "reduce":
        agg[("reduce after", prev, r["Grid_Size"] if "Grid_Size" in r else "")].append(d)
    else:
        agg[(short, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))].append(d); prev = short
for k, v in agg.items():
    v = v[len(v) // 2:]
    print(k, "n=%d avg %.1f us min %.1f max %.1f" % (len(v), sum(v) / len(v), min(v), max(v)))
