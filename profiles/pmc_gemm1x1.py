#!/usr/bin/env python3
"""Runs ONLY hdrsky_gemm1x1_bf16 on the shape bench.py's distortion-aware roofline object times (G [8,32,128,1152] bf16 x a 128-filter
image, fp32 output + InstanceNorm partials) 24 times over 4 different operands, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in
separate runs, MI355X_MICROARCH.md "HBM")."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
pw = K.PackedConv(torch.randn(3, 3, 128, 128, device=dev) / 34.0, False).as_1x1()
bias = torch.zeros(128, device=dev)
Gs = [torch.randn(8, 32, 128, 1152, device=dev).to(torch.bfloat16) for _ in range(4)]
for i in range(24):
    K.gemm1x1(Gs[i % 4], pw, bias, want_stats=True)
torch.cuda.synchronize()
print("done")
