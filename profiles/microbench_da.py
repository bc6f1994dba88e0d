#!/usr/bin/env python3
"""Distortion-aware 3x3 conv vs the plain 3x3 conv on the res-block shape, low-res batch 32 and hi-res batch 8."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdrsky_amd as hs
K, ops_da = (importlib.import_module(hs.__name__ + "." + m) for m in ("kernels", "distortion_aware_ops"))
dev = torch.device("cuda:0")


def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (B, H, W, C, F) in ((32, 8, 32, 128, 128), (8, 32, 128, 128, 128), (32, 16, 64, 64, 64)):
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(3, 3, C, F, device=dev) / (9 * C) ** 0.5
    pw = K.PackedConv(w, False); bias = torch.zeros(F, device=dev)
    us_p = timeit(lambda: K.conv2d(x, pw, bias))
    da = ops_da.conv2d(F, 3, compute=K.BF16); da(x)
    us_d = timeit(lambda: da(x))
    flop = 2.0 * B * H * W * 9 * C * F
    print("[%d,%d,%d,%d]->%d  plain %.1f us (%.0f TF)   distortion-aware %.1f us (%.0f TF)" %
          (B, H, W, C, F, us_p, flop / us_p / 1e6, us_d, flop / us_d / 1e6), flush=True)
