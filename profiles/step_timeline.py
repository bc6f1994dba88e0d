#!/usr/bin/env python3
"""Kernel-by-kernel timeline of ONE bench-mode training step from a rocprofv3 --kernel-trace CSV of
`bench.py --workload train --steps-only`: step_timeline.py DIR [queue rank]  - the step is the last complete run between two
launches of the marker kernel (rmsprop2_kernel: both optimizers' conv-side update, ONE launch per bench-mode step at the end
of `apply`; round 4's marker `rmsprop_kernel` is launched by the fp32-class step only, which is how r04_step_timeline.txt came
to describe a BF16X3 step).  The chosen window must not hold a fp32-class (PRECISE = true) instantiation: asserted.
Kernels are listed per HIP stream (queue) with their start offset, duration and the gap to the previous kernel of the queue."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:78]
import re
def precise(name):
    m = re.search(r"conv_igemm_kernel<([^>]*)>", name)
    if m:
        a = [t.strip() for t in m.group(1).split(",")]
        return len(a) > 6 and a[6] == "true"
    m = re.search(r"fc_mfma_kernel<([^>]*)>", name)
    if m:
        return [t.strip() for t in m.group(1).split(",")][1] == "true"
    return "conv_wgrad_kernel<" in name or "rmsprop_kernel(" in name
marks = [i for i, r in enumerate(rows) if "rmsprop2_kernel" in r["Kernel_Name"]]
assert len(marks) >= 4, "no rmsprop2_kernel launches: not a trace of bench-mode steps"
# one marker per step.  A step ends with a join of its three streams, so its successor's first kernel is the first one
# behind the marker that starts after EVERYTHING before it has ended (the Dense update on another stream may still run
# when rmsprop2 starts, the filter re-pack follows it)
def step_start(m):
    end = max(r["e"] for r in rows[max(0, m - 40):m + 1])
    for i in range(m + 1, min(len(rows), m + 40)):
        if rows[i]["s"] >= end:
            return i
        end = max(end, rows[i]["e"])
    raise SystemExit("no stream join found behind marker %d" % m)
a, b = step_start(marks[-3]), step_start(marks[-2])
step = rows[a:b]
bad = [r["Kernel_Name"] for r in step if precise(r["Kernel_Name"])]
assert not bad, "fp32-class instantiation inside the chosen step: %s" % bad[0]
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
