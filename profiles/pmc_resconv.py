#!/usr/bin/env python3
"""Runs ONLY the roofline kernel of bench.py - the sample-resident res-block launch (conv 3x3 128->128 + InstanceNorm + leaky
on [32,8,32,128] bf16, training-forward form: bf16 activation + xhat + rstd out) - 40 times, for rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE / MFMA counters in separate runs, MI355X_MICROARCH.md "HBM").  x cycles through 8 different input
tensors so the reads cannot all be served by a warm L2."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
w = torch.randn(3, 3, 128, 128, device=dev) / 34.0
pw = K.PackedConv(w, precise=False)
gamma, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)
xs = [torch.randn(32, 8, 32, 128, device=dev).to(torch.bfloat16) for _ in range(8)]
for i in range(40):
    K.resconv_fwd(xs[i % 8], pw, None, gamma, beta, 0.1, save=True)
torch.cuda.synchronize()
print("done")
