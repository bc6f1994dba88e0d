#!/usr/bin/env python3
"""Microbenchmark of hdrsky_norm_act_bwd alone on the chip (hipGraph of N launches, HIP events): the one-launch register-resident
form (round 5, norm_act_bwd1_kernel) against the sliced reduce + apply pair (HDRSKY_NAB_ONE=0), per shape of the 32x128 network,
with the incoming gradient / output / x storage the training step uses (bf16 dy, bf16 dx) and in fp32."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
HK = importlib.import_module(PKG + ".hooks")
dev = torch.device("cuda:0")
B, iters = 32, 50


def timeit(fn):
    fn()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


SHAPES = [(32, 128, 32, 0), (16, 64, 64, 0), (8, 32, 128, 0), (32, 128, 32, 1), (16, 64, 64, 1), (8, 32, 128, 1)]
if "--hires" in sys.argv:      # the 128x512 network at batch 8
    B = 8
    SHAPES = [(128, 512, 32, 0), (64, 256, 64, 0), (32, 128, 128, 0), (128, 512, 32, 1), (64, 256, 64, 1)]
for (H, W, C, pooled) in SHAPES:
    x = torch.randn(B, H, W, C, device=dev)
    nparts = 8
    part = torch.rand(B, nparts, 2, C, device=dev) * (H * W / nparts); part[:, :, 1] += H * W / nparts
    st = K.Stats(part, nparts, H * W)
    g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.randn(B, H // 2 if pooled else H, W // 2 if pooled else W, C, device=dev)
    row = []
    for dy16, ob in ((True, True), (False, False)):
        dyi = dy.to(torch.bfloat16) if dy16 else dy
        us = {}
        for one in ("1", "0"):
            os.environ["HDRSKY_NAB_ONE"] = one; HK.reload()
            us[one] = timeit(lambda: K.norm_act_bwd(x, st, g, bt, 0.1, dyi, bool(pooled), out_bf16=ob))
        os.environ.pop("HDRSKY_NAB_ONE"); HK.reload()
        row.append("%s: one launch %6.2f us | sliced %6.2f us" % ("bf16 dy/dx" if dy16 else "fp32 dy/dx", us["1"], us["0"]))
    mb = B * H * W * C * (4 + 2 + 2 / (4 if pooled else 1)) / 1e6      # x fp32 once + dx bf16 + dy bf16: the one-read floor
    print("[%d,%d,%d,%d] pooled=%d  %s   (floor %.1f MB = %.1f us at 8 TB/s)" % (B, H, W, C, pooled, "   ".join(row), mb, mb / 8.0), flush=True)
