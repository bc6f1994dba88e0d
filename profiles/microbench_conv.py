#!/usr/bin/env python3
"""Per-layer microbenchmark of hdrsky_conv2d (forward implicit GEMM, bf16 MFMA) on the layer shapes of the training step
at batch 32: hipGraph of N launches, HIP events.  usage: python profiles/microbench_conv.py [--only vgg] [--tile "..."]"""
import argparse, importlib, os, sys
import torch

def _hook(name, value):
    """Set / clear a HDRSKY_* variable and make the package + library read it (they read the environment once:
    hooks.py, csrc/hooks.h; tuning hooks need the HDRSKY_EXPERIMENTS=1 gate)."""
    import importlib, os, sys
    os.environ["HDRSKY_EXPERIMENTS"] = "1"
    if value is None: os.environ.pop(name, None)
    else: os.environ[name] = str(value)
    mods = [m for n, m in sys.modules.items() if n.endswith("_amd.hooks")]
    if mods: mods[0].reload()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")

# name, H, W, Cin, Cout, k, stride, launches per training step (fwd + dgrad)
LAYERS = [
    ("vgg1_1 3->64 @32x128", 32, 128, 3, 64, 3, 1, 2),
    ("vgg1_2 64->64 @32x128", 32, 128, 64, 64, 3, 1, 3),
    ("vgg2_1 64->128 @16x64", 16, 64, 64, 128, 3, 1, 3),
    ("vgg2_2 128->128 @16x64", 16, 64, 128, 128, 3, 1, 3),
    ("vgg3_1 128->256 @8x32", 8, 32, 128, 256, 3, 1, 3),
    ("vgg3_2 256->256 @8x32", 8, 32, 256, 256, 3, 1, 6),
    ("res 128->128 @8x32", 8, 32, 128, 128, 3, 1, 42),
    ("l2b 64->64 @16x64", 16, 64, 64, 64, 3, 1, 3),
    ("l1b 7x7 32->32 @32x128", 32, 128, 32, 32, 7, 1, 2),
    ("conv2_d 3x3s2 32->64", 32, 128, 32, 64, 3, 2, 2),
    ("d2 4x4s2 64->128", 32 // 2, 128 // 2, 64, 128, 4, 2, 6),
    ("dec1 7x7 32->3 @32x128", 32, 128, 32, 3, 7, 1, 2),
    ("d3 4x4s2 128->256 @8x32", 8, 32, 128, 256, 4, 2, 8),
    ("d4 4x4 256->512 @4x16", 4, 16, 256, 512, 4, 1, 8),
    ("vgg1_1 dgrad 64->3 @32x128", 32, 128, 64, 3, 3, 1, 2),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--tiles", default="", help="semicolon-separated HDRSKY_TILE values to compare with the table")
    ap.add_argument("--scale", type=int, default=1, help="multiply the map sizes (4: the 128x512 workload, use --batch 8)")
    ap.add_argument("--bf16", action="store_true", help="bf16 activations in and out + ReLU epilogue (the VGG16 chain of the step)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    cfgs = [""] + [t for t in args.tiles.split(";") if t]
    tot = [0.0] * len(cfgs)
    for (name, H, W, Cin, Cout, k, stride, cnt) in LAYERS:
        if args.only and args.only not in name:
            continue
        H, W = H * args.scale, W * args.scale
        x = torch.randn(B, H, W, Cin, device=dev)
        kw = {}
        if args.bf16 and Cin >= 32:
            x = x.to(torch.bfloat16); kw = dict(out_bf16=True, out_slope=0.0)
        w = torch.randn(k, k, Cin, Cout, device=dev) / (k * k * Cin) ** 0.5
        pw = K.PackedConv(w, False); bias = torch.zeros(Cout, device=dev)
        res = []
        for ci, t in enumerate(cfgs):
            if t: _hook("HDRSKY_TILE", t)
            else: _hook("HDRSKY_TILE", None)
            try:
                y = K.conv2d(x, pw, bias, stride=stride, **kw)[0]
            except Exception as e:
                res.append("%s n/a" % t); continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(args.iters):
                    K.conv2d(x, pw, bias, stride=stride, **kw)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            flop = 2.0 * y.numel() * k * k * Cin
            tot[ci] += us * cnt
            res.append("%s %7.2f us %6.1f TF" % (t or "table", us, flop / us / 1e6))
        print("%-26s x%-2d | %s" % (name, cnt, " | ".join(res)), flush=True)
    print("per-step total (us): " + " | ".join("%s %.0f" % (c or "table", t) for c, t in zip(cfgs, tot)))

if __name__ == "__main__":
    main()
