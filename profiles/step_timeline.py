#!/usr/bin/env python3
"""Kernel-by-kernel timeline of ONE training step from a rocprofv3 --kernel-trace CSV of `bench.py --workload train`:
step_timeline.py DIR [queue rank]  - the step is the last complete run between two launches of the marker kernel
(rmsprop_kernel, the generator's update at the end of `apply`); kernels are listed per HIP stream (queue) with their start
offset, duration and the gap to the previous kernel of the same queue."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:78]
marks = [i for i, r in enumerate(rows) if "rmsprop_kernel" in r["Kernel_Name"]]
# marker launches come in bursts (several per apply); a step boundary = a gap of > 1 ms between marker launches
bounds = [marks[0]] + [marks[k] for k in range(1, len(marks)) if rows[marks[k]]["s"] - rows[marks[k - 1]]["s"] > 1_000_000]
lo, hi = bounds[-3], bounds[-2]
# the step runs from just after the last marker burst of step n-1 to the last marker of step n
last_of = lambda b: max(i for i in marks if rows[i]["s"] - rows[b]["s"] < 500_000 and i >= b)
a, b = last_of(lo) + 1, last_of(hi) + 1
step = rows[a:b]
t0 = step[0]["s"]
qs = collections.OrderedDict()
for r in step:
    qs.setdefault(r["Queue_Id"], []).append(r)
print("step: %d kernels, %.1f us, queues %s" % (len(step), (step[-1]["e"] - t0) / 1e3, {q: len(v) for q, v in qs.items()}))
only = sys.argv[2] if len(sys.argv) > 2 else None
for qi, (q, v) in enumerate(qs.items()):
    if only is not None and str(qi) != only:
        continue
    busy = sum(r["e"] - r["s"] for r in v) / 1e3
    print("== queue %s (#%d): %d kernels, busy %.1f us" % (q, qi, len(v), busy))
    prev = None
    for r in v:
        gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
        print("  %8.1f  %7.1f us  gap %6.1f  grid %-8s %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, gap,
                                                              r.get("Grid_Size_X") or r.get("Grid_Size"), short(r["Kernel_Name"])))
        prev = r["e"]
