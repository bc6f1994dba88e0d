#!/bin/bash
P=$(ls -d *_amd)
for v in base direct; do cp ab/$v.so $P/libhdrsky.so; python3 profiles/dbg_epi_calls.py /tmp/calls_$v.pkl 2>&1 | grep -v amdgpu.ids; done
cp ab/direct.so $P/libhdrsky.so
python3 - <<'PY'
import pickle, numpy as np
a, b = pickle.load(open("/tmp/calls_base.pkl", "rb")), pickle.load(open("/tmp/calls_direct.pkl", "rb"))
print(len(a), len(b))
for i, (ca, cb) in enumerate(zip(a, b)):
    dx = np.abs(ca[4] - cb[4]).max(); dy = np.abs(ca[2] - cb[2]).max()
    ds = 0.0 if ca[3] is None else np.abs(ca[3] - cb[3]).max()
    print("%3d %-34s %-70s |dx| %.2e |dy| %.2e (max|y| %.2e) |dstats| %.2e" % (i, str(ca[0])[:34], ca[1], dx, dy, np.abs(ca[2]).max(), ds))
PY
