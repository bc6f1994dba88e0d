#!/usr/bin/env python3
"""Median duration per kernel name (and grid size) from rocprofv3 --kernel-trace CSV files: trace_split.py DIR [substring]"""
import csv, glob, sys
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in sorted(glob.glob(d + "/*kernel_trace.csv")):
    agg = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if sub and sub not in n:
            continue
        key = (n.replace("(anonymous namespace)::", "")[:70], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("LDS_Block_Size"))
        agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("==", f.split("/")[-1])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        print("   %-72s grid=%-7s lds=%-6s n=%-3d med=%7.1f us  min=%7.1f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
