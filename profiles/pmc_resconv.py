#!/usr/bin/env python3
"""Runs ONLY the roofline kernel of bench.py (res-block conv 3x3 128->128 on [32,8,32,128], bf16) 40 times, for
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs, MI355X_MICROARCH.md "HBM").  x cycles through 64
different input tensors (256 MB) so the reads cannot all be served by a warm L2."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
w = torch.randn(3, 3, 128, 128, device=dev) / 34.0
pw = K.PackedConv(w, precise=False); bias = torch.zeros(128, device=dev)
xs = [torch.randn(32, 8, 32, 128, device=dev) for _ in range(8)]
y = torch.empty_like(xs[0])
for i in range(40):
    K.conv2d(xs[i % 8], pw, bias, want_stats=True, compute=K.BF16, out=y)
torch.cuda.synchronize()
print("done")
