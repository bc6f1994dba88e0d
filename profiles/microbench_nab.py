#!/usr/bin/env python3
"""Microbenchmark of hdrsky_norm_act_bwd (hipGraph of N launches, HIP events).  HDRSKY_NAB_TARGET selects the split."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
B, iters = 32, 50
for (H, W, C, pooled) in [(32, 128, 32, 0), (16, 64, 64, 0), (8, 32, 128, 0), (32, 128, 32, 1), (16, 64, 64, 1), (8, 32, 128, 1)]:
    x = torch.randn(B, H, W, C, device=dev)
    nparts = 8
    part = torch.rand(B, nparts, 2, C, device=dev) * (H * W / nparts); part[:, :, 1] += H * W / nparts
    st = K.Stats(part, nparts, H * W)
    g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.randn(B, H // 2 if pooled else H, W // 2 if pooled else W, C, device=dev)
    K.norm_act_bwd(x, st, g, bt, 0.1, dy, bool(pooled))
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            K.norm_act_bwd(x, st, g, bt, 0.1, dy, bool(pooled))
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    mb = (x.numel() * 2 + dy.numel()) * 4 / 1e6
    print("[%d,%d,%d,%d] pooled=%d  %7.2f us  (%.1f MB min traffic, %.2f TB/s)" % (B, H, W, C, pooled, us, mb, mb / us / 1e6 * 1e6 / 1e6), flush=True)
