#!/usr/bin/env python3
"""fc1 of the sun-pose net (8192 -> 4096, M = 32 rows) forward and data gradient alone: us per launch and GB/s of weight stream
(hipGraph of 20 launches, HIP events; the weights - 67 MB per image - are re-read from HBM every launch)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
w = torch.randn(8192, 4096, device=dev) * 0.01
fc = K.PackedFC(w, precise=False)
x = torch.randn(32, 8192, device=dev)
dy = torch.randn(32, 4096, device=dev)


def gtime(fn, iters=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, fn in (("forward", lambda: K.fc_fwd(x, fc, K.BF16)), ("data gradient", lambda: K.fc_dgrad(dy, fc, K.BF16))):
    t = gtime(fn)
    print("fc1 %-14s %6.1f us  %7.1f GB/s" % (name, t, 8192 * 4096 * 2 / t / 1e3))
