#!/usr/bin/env python3
"""Per-layer microbenchmark of hdrsky_conv2d_fwd (hipGraph of N launches, HIP events).
usage: python profiles/microbench_conv.py [--batch 32] [--tiles "2,2,2,2,32;2,2,2,1,32"]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels"); L = importlib.import_module(PKG + "._lib")

# name, H, W, Cin, Cout, k, stride, upsample, xf(PARTIALS?), stats
LAYERS = [
    ("g.conv1_d 7x7 3->32", 32, 128, 3, 32, 7, 1, 1, False, True),
    ("g.conv2_d 3x3s2 32->64", 32, 128, 32, 64, 3, 2, 1, True, True),
    ("g.conv3_d 3x3s2 64->128", 16, 64, 64, 128, 3, 2, 1, True, True),
    ("g.res 3x3 128->128", 8, 32, 128, 128, 3, 1, 1, True, True),
    ("g.dec3 up3x3 128->64", 8, 32, 128, 64, 3, 1, 2, False, True),
    ("g.dec2 up3x3 64->32", 16, 64, 64, 32, 3, 1, 2, True, True),
    ("g.dec1 7x7 32->3", 32, 128, 32, 3, 7, 1, 1, True, False),
    ("s.l1b 7x7 32->32", 32, 128, 32, 32, 7, 1, 1, True, True),
    ("s.l2b 3x3 64->64", 16, 64, 64, 64, 3, 1, 1, True, True),
    ("d2 4x4s2 64->128", 16, 64, 64, 128, 4, 2, 1, False, False),
    ("d3 4x4s2 128->256", 8, 32, 128, 256, 4, 2, 1, False, False),
    ("d4 4x4 256->512", 4, 16, 256, 512, 4, 1, 1, False, False),
    ("tiny 3x3 32->32 @8x32", 8, 32, 32, 32, 3, 1, 1, False, False),
    ("tiny 1x1 32->32 @8x32", 8, 32, 32, 32, 1, 1, 1, False, False),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--tiles", default="")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    tiles = [t for t in args.tiles.split(";") if t] or [""]
    for (name, H, W, Cin, Cout, k, stride, up, xfm, stats) in LAYERS:
        if args.only and args.only not in name:
            continue
        x = torch.randn(B, H, W, Cin, device=dev)
        w = torch.randn(k, k, Cin, Cout, device=dev) / (k * k * Cin) ** 0.5
        pw = K.PackedConv(w, precise=False)
        bias = torch.zeros(Cout, device=dev)
        xf = None
        if xfm:
            # partials of x itself via a 1x1 conv so the PARTIALS prologue has real data
            nparts = 8
            part = torch.rand(B, nparts, 2, Cin, device=dev) * (H * W / nparts)
            part[:, :, 1] += (H * W / nparts)
            st = K.Stats(part, nparts, H * W)
            xf = K.InXf(mode=L.IN_PARTIALS, slope=0.1, stats=st, gamma=torch.ones(Cin, device=dev), beta=torch.zeros(Cin, device=dev))
        d = K.conv_desc(B, H, W, Cin, Cout, k, k, stride, True, up)
        flop = 2.0 * B * d.Ho * d.Wo * k * k * Cin * Cout
        res = []
        for t in tiles:
            if t: os.environ["HDRSKY_TILE"] = t
            else: os.environ.pop("HDRSKY_TILE", None)
            try:
                y, _ = K.conv2d(x, pw, bias, stride=stride, upsample=up, xf=xf, want_stats=stats)
            except Exception as e:
                res.append("%s: n/a" % t); continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(args.iters):
                    K.conv2d(x, pw, bias, stride=stride, upsample=up, xf=xf, want_stats=stats, out=y)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            res.append("%s: %6.2f us %6.1f TF" % (t or "auto", us, flop / us / 1e6))
        print("%-26s %6.2f GF | %s" % (name, flop / 1e9, " | ".join(res)), flush=True)

if __name__ == "__main__":
    main()
