#!/usr/bin/env python3
"""Times hdrsky_jpeg_roundtrip (two launches) with HIP events over a hipGraph of N calls; prints the HBM rate.
usage: python profiles/microbench_jpeg.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
for (b, h, w) in ((32, 32, 128), (256, 32, 128), (32, 64, 256), (1024, 64, 256)):
    x = (torch.randint(0, 256, (b, h, w, 3), device=dev).float() / 255.0).contiguous()
    out = torch.empty_like(x)
    K.jpeg_roundtrip(x, out=out); torch.cuda.synchronize()
    iters = 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            K.jpeg_roundtrip(x, out=out)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    nbytes = b * h * w * (12 + 1.5 + 1.5 + 12)
    print("B=%4d %3dx%3d  %8.2f us  %7.1f GB/s algorithmic  %9.0f images/s" % (b, h, w, us, nbytes / us / 1e3, b / us * 1e6), flush=True)
