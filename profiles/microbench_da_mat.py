#!/usr/bin/env python3
"""Distortion-aware 3x3 layer, forward / data gradient / kernel gradient alone, on the written gathered operand
(HDRSKY_DA_MAT=1: hdrsky_da_gather_bf16 + generic 1x1 conv / weight gradient) against the fused kernels (HDRSKY_DA_MAT=0), for
the layer shapes of the 32x128 and 128x512 networks.  hipGraph of N launches, HIP events.  The policy of kernels.da_mat_ok /
kernels.da_conv2d comes from this table (profiles/r04_da_mat_ab.txt)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
HK = importlib.import_module(PKG + ".hooks")
dev = torch.device("cuda:0")
K.DA_MAT_MIN_PIXELS = {"fwd": 0, "dgrad": 0, "wgrad": 0}          # the table decides: both paths everywhere

SHAPES = [  # name, B, H, W, C, F
    ("res 32x128 net", 32, 8, 32, 128, 128),
    ("dec conv3 32x128 net", 32, 16, 64, 128, 64),
    ("dec conv2 32x128 net", 32, 32, 128, 64, 32),
    ("sun l2 conv2", 32, 16, 64, 64, 64),
    ("res 128x512 net", 8, 32, 128, 128, 128),
    ("dec conv3 128x512 net", 8, 64, 256, 128, 64),
    ("dec conv2 128x512 net", 8, 128, 512, 64, 32),
]


def gtime(fn, iters=10):
    for _ in range(2):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, B, H, W, C, F in SHAPES:
    x = torch.randn(B, H, W, C, device=dev)
    kern = torch.randn(9 * C, F, device=dev) / (9 * C) ** 0.5
    bias = torch.zeros(F, device=dev)
    pw = K.PackedConv(kern.view(3, 3, C, F), precise=False)
    pwT = K.PackedConv(kern.view(3, 3, C, F), precise=False, transpose_flip=True)
    offs = K.da_offsets_device(H, W, 3, device=dev)
    table = K.da_transpose_table(H, W, 3, device=dev)
    dw, db = torch.zeros(9 * C, F, device=dev), torch.zeros(F, device=dev)
    row = []
    for mat in ("1", "0"):
        os.environ["HDRSKY_DA_MAT"] = mat; HK.reload()
        op = K.Operand()
        dy = torch.randn(B, H, W, F, device=dev)
        if mat == "1":
            dy = dy.to(torch.bfloat16)
        def fwd():      # (the operand goes to the caller's handle, which the kernel gradient below reads)
            return K.da_conv2d(x, pw, bias, offs, K.BF16, want_stats=True, train=True, operand=op)
        t_f = gtime(fwd)
        t_d = gtime(lambda: K.da_conv2d_dgrad(dy, pwT, table, 3, K.BF16))
        t_w = gtime(lambda: K.conv2d_wgrad_multi([K.da_wgrad_job(x, dy, 3, offs, dw, db, K.BF16, operand=op)]))       # (operand kept by the forward when written)
        t_g = gtime(lambda: K.da_gather_bf16(x, offs, ksize=3)) if mat == "1" else 0.0
        row.append("%s: fwd %7.1f  dgrad %7.1f  wgrad %7.1f%s" % ("written" if mat == "1" else "fused  ", t_f, t_d, t_w,
                                                                 "  (gather alone %6.1f)" % t_g if mat == "1" else ""))
    gf = 2.0 * B * H * W * 9 * C * F / 1e9
    print("%-24s B=%-2d %3dx%-3d %3d->%-3d %5.1f GFLOP | %s | %s" % (name, B, H, W, C, F, gf, row[0], row[1]), flush=True)
