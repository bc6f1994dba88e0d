#!/usr/bin/env python3
"""Res-block conv 3x3 128->128 on [B,8,32,128]: the generic implicit-GEMM launch (fp32 activations, consumer-side
InstanceNorm) against the sample-resident launch (bf16 activations, InstanceNorm in the epilogue), hipGraph of N launches
between HIP events.  usage: python profiles/microbench_resconv.py [--batch 32] [--iters 50]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev, B, C = torch.device("cuda:0"), args.batch, 128
    x = torch.randn(B, 8, 32, C, device=dev)
    w = torch.randn(3, 3, C, C, device=dev) / (9 * C) ** 0.5
    pw, pwT = K.PackedConv(w, False), K.PackedConv(w, False, transpose_flip=True)
    bias, gamma, beta = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    xb = x.to(torch.bfloat16)
    flop = 2.0 * B * 256 * 9 * C * C
    _, st = K.conv2d(x, pw, bias, want_stats=True)
    xf = K.InXf(mode=1 + 1, slope=0.1, stats=st, gamma=gamma, beta=beta)
    rows = [
        ("generic conv (IN partials in, stats out)", lambda: K.conv2d(x, pw, bias, want_stats=True, xf=xf)),
        ("resconv fwd act (bf16 out)", lambda: K.resconv_fwd(xb, pw, bias, gamma, beta, 0.1)),
        ("resconv fwd act + save (training)", lambda: K.resconv_fwd(xb, pw, bias, gamma, beta, 0.1, save=True)),
        ("resconv fwd res (f32 + bf16 + save)", lambda: K.resconv_fwd(xb, pw, bias, gamma, beta, 1.0, residual=x, want_f32=True, save=True)),
    ]
    o = K.resconv_fwd(xb, pw, bias, gamma, beta, 0.1, save=True)
    dgb = torch.zeros(B, 2, C, device=dev)
    nd = dict(xhat=o["xhat"], inv=o["inv"], gamma=gamma, beta=beta, slope=0.1, dgb=dgb)
    rows += [
        ("resconv bwd (dgrad + IN/leaky backward)", lambda: K.resconv_bwd(xb, pwT, norm=nd)),
        ("resconv bwd (dgrad + skip + IN backward)", lambda: K.resconv_bwd(xb, pwT, skip=x, norm=nd, want_f32=True)),
        ("resconv no-conv (IN backward only)", lambda: K.resconv_bwd(None, None, skip=x, norm=nd)),
    ]
    for name, fn in rows:
        us = timed(fn, args.iters)
        print("%-44s %7.2f us  %7.1f TFLOP/s  frac of 2.5 PF %.3f" % (name, us, flop / us / 1e6, flop / us / 1e6 / 2500.0), flush=True)


if __name__ == "__main__":
    main()
