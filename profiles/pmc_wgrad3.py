#!/usr/bin/env python3
"""Runs the round-3 kernels 20 times each for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md "HBM" prescribes): conv_wgrad3_kernel + its reduce on the 7x7 3->32 stem and the 7x7 32->3 tail (batch 32,
32x128), conv_wgrad2_kernel + reduce on the 12 res-block layers, and the one-launch DoG term.  Inputs cycle through 4 tensor
sets so that the reads are not all served by a warm L2."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")


def job(B, H, W, Cin, Cout, k, bf16x, bf16y):
    x = torch.randn(B, H, W, Cin, device=dev)
    dy = torch.randn(B, H, W, Cout, device=dev)
    if bf16x: x = x.to(torch.bfloat16)
    if bf16y: dy = dy.to(torch.bfloat16)
    return K.wgrad_job(x, dy, k, k, torch.zeros(k, k, Cin, Cout, device=dev), torch.zeros(Cout, device=dev), compute=K.BF16)


stems = [job(32, 32, 128, 3, 32, 7, False, True) for _ in range(4)]
tails = [job(32, 32, 128, 32, 3, 7, True, False) for _ in range(4)]
res = [[job(32, 8, 32, 128, 128, 3, True, True) for _ in range(12)] for _ in range(2)]
ys = [torch.rand(32, 32, 128, 3, device=dev) * 3 for _ in range(4)]
ts = [torch.rand(32, 32, 128, 3, device=dev) * 3 for _ in range(4)]
slot, dyo = torch.zeros(1, device=dev), torch.zeros(32, 32, 128, 3, device=dev)
for i in range(20):
    K.conv2d_wgrad_multi([stems[i % 4]])
for i in range(20):
    K.conv2d_wgrad_multi([tails[i % 4]])
for i in range(20):
    K.conv2d_wgrad_multi(res[i % 2])
for i in range(20):
    K.dog_loss(ys[i % 4], ts[i % 4], 1000.0, slot, dyo)
torch.cuda.synchronize()
print("done")
