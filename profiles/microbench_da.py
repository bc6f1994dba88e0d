#!/usr/bin/env python3
"""Distortion-aware conv, forward and data gradient: the plain conv of the same shape, the global-memory gather
(HDRSKY_DA_REGION=0) and the LDS-region variant, on the layer shapes of the model (batch 32 at 32x128)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdrsky_amd as hs
K = importlib.import_module(hs.__name__ + ".kernels")
HK = importlib.import_module(hs.__name__ + ".hooks")      # HDRSKY_* variables are read once: reload() after every change
dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def both(fn):
    out = []
    for mode in ("0", "2"):
        os.environ["HDRSKY_DA_REGION"] = mode; HK.reload()
        try:
            out.append(timeit(fn))
        except Exception:
            out.append(float("nan"))
    os.environ.pop("HDRSKY_DA_REGION"); HK.reload()
    return out


for (B, H, W, C, F, k) in ((32, 8, 32, 128, 128, 3), (32, 32, 128, 32, 32, 7), (32, 16, 64, 64, 64, 5), (32, 16, 64, 32, 64, 5),
                           (32, 8, 32, 64, 128, 3), (32, 4, 16, 256, 256, 3), (32, 16, 64, 128, 64, 3), (32, 32, 128, 64, 32, 3),
                           (8, 32, 128, 128, 128, 3)):
    x = torch.randn(B, H, W, C, device=dev); dy = torch.randn(B, H, W, F, device=dev)
    w = torch.randn(k, k, C, F, device=dev) / (k * k * C) ** 0.5
    pw = K.PackedConv(w, False); pwT = K.PackedConv(w, False, transpose_flip=True); bias = torch.zeros(F, device=dev)
    offs = K.da_offsets_device(H, W, k, 1, True, dev); table = K.da_transpose_table(H, W, k, 1, True, dev)
    us_p = timeit(lambda: K.conv2d(x, pw, bias))
    f_g, f_r = both(lambda: K.da_conv2d(x, pw, bias, offs, K.BF16))
    d_g, d_r = both(lambda: K.da_conv2d_dgrad(dy, pwT, table, k, K.BF16))
    flop = 2.0 * B * H * W * k * k * C * F
    print("%dx%d [%d,%d,%d,%d]->%d  plain %.1f us (%.0f TF) | fwd: global %.1f, region %.1f us (%.0f TF) | dgrad: global %.1f, region %.1f us (rows %d / %d)" %
          (k, k, B, H, W, C, F, us_p, flop / us_p / 1e6, f_g, f_r, flop / f_r / 1e6, d_g, d_r, offs.da_rows[1][0], table[0].da_rows[1][0]), flush=True)
