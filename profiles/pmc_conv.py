#!/usr/bin/env python3
"""Runs a handful of representative conv_igemm launches alone, 12 times each, for rocprofv3 --pmc passes (pmc_conv.sh): where
do the wave cycles of the step's dominant kernel family go - parked (s_waitcnt / barrier), issue-stalled, LDS, matrix core?
Each case prints its grid size so that the counter rows can be told apart (the kernel name alone is the tile)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
# (name, B, H, W, Cin, Cout, k, bf16 in/out)
CASES = [("vgg 64->64 @32x128 B32", 32, 32, 128, 64, 64, 3, True), ("vgg 128->128 @16x64 B32", 32, 16, 64, 128, 128, 3, True),
         ("vgg 256->256 @8x32 B32", 32, 8, 32, 256, 256, 3, True), ("dec 64->32 @32x128 B32 f32+stats", 32, 32, 128, 64, 32, 3, False),
         ("res 128->128 @32x128 B8 f32+stats (128x512 net)", 8, 32, 128, 128, 128, 3, False)]
for name, B, H, W, C, F, k, bf in CASES:
    xs = [torch.randn(B, H, W, C, device=dev) for _ in range(4)]
    if bf:
        xs = [x.to(torch.bfloat16) for x in xs]
    pw = K.PackedConv(torch.randn(k, k, C, F, device=dev) * 0.05, precise=False)
    bias = torch.zeros(F, device=dev)
    for i in range(12):
        if bf:
            K.conv2d(xs[i % 4], pw, bias, compute=K.BF16, out_slope=0.0, out_bf16=True)
        else:
            K.conv2d(xs[i % 4], pw, bias, compute=K.BF16, want_stats=True)
    torch.cuda.synchronize()
    print("%s: %.3f GFLOP per launch" % (name, 2e-9 * B * H * W * k * k * C * F), flush=True)
print("done")
