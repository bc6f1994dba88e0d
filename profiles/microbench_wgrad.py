#!/usr/bin/env python3
"""Per-layer microbenchmark of hdrsky_conv2d_wgrad (hipGraph of N launches incl. the zero-fill it needs, HIP events).
usage: python profiles/microbench_wgrad.py [--batch 32] [--cfg "4,4,32,128;2,2,32,256"]   (HDRSKY_WGRAD hook values)"""
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
K = importlib.import_module(PKG + ".kernels"); L = importlib.import_module(PKG + "._lib")

# name, H, W, Cin, Cout, k, stride, upsample, count per training step
LAYERS = [
    ("res 3x3 128->128 @8x32", 8, 32, 128, 128, 3, 1, 1, 14),
    ("l2b 3x3 64->64 @16x64", 16, 64, 64, 64, 3, 1, 1, 1),
    ("l2a 3x3 32->64 @16x64", 16, 64, 32, 64, 3, 1, 1, 1),
    ("conv2_d 3x3s2 32->64", 32, 128, 32, 64, 3, 2, 1, 1),
    ("conv3_d 3x3s2 64->128", 16, 64, 64, 128, 3, 2, 1, 2),
    ("dec3 up3x3 128->64", 8, 32, 128, 64, 3, 1, 2, 2),
    ("dec2 up3x3 64->32", 16, 64, 64, 32, 3, 1, 2, 2),
    ("dec1 7x7 32->3", 32, 128, 32, 3, 7, 1, 1, 2),
    ("l1b 7x7 32->32", 32, 128, 32, 32, 7, 1, 1, 1),
    ("conv1 7x7 3->32", 32, 128, 3, 32, 7, 1, 1, 2),
    ("d1 4x4s2 6->64", 32, 128, 6, 64, 4, 2, 1, 3),
    ("d2 4x4s2 64->128", 16, 64, 64, 128, 4, 2, 1, 3),
    ("d3 4x4s2 128->256", 8, 32, 128, 256, 4, 2, 1, 3),
    ("d4 4x4 256->512", 4, 16, 256, 512, 4, 1, 1, 3),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--cfg", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--group", type=int, default=0, help="N copies of each layer in ONE multi launch (per-layer time)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    cfgs = args.cfg.split(";")
    total = [0.0] * len(cfgs)
    for (name, H, W, Cin, Cout, k, stride, up, cnt) in LAYERS:
        if args.only and args.only not in name:
            continue
        x = torch.randn(B, H, W, Cin, device=dev)
        d = K.conv_desc(B, H, W, Cin, Cout, k, k, stride, True, up)
        dy = torch.randn(B, d.Ho, d.Wo, Cout, device=dev)
        flop = 2.0 * B * d.Ho * d.Wo * k * k * Cin * Cout
        res = []
        for ci, t in enumerate(cfgs):
            if t: _hook("HDRSKY_WGRAD", t)
            else: _hook("HDRSKY_WGRAD", None)
            try:
                dw, db = K.conv2d_wgrad(x, dy, k, k, stride, True, up)
                if args.group:
                    xs = [torch.randn_like(x) for _ in range(args.group)]
                    dys = [torch.randn_like(dy) for _ in range(args.group)]
                    jobs = [K.wgrad_job(xs[i], dys[i], k, k, torch.zeros_like(dw), torch.zeros_like(db), stride=stride,
                                        upsample=up) for i in range(args.group)]
                    K.conv2d_wgrad_multi(jobs)
            except Exception as e:
                res.append("%s: n/a" % t); continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(args.iters):
                    if args.group:
                        K.conv2d_wgrad_multi(jobs)
                    else:
                        K.conv2d_wgrad(x, dy, k, k, stride, True, up, dw=dw, db=db)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters / max(1, args.group)
            total[ci] += us * cnt
            res.append("%s: %6.2f us %5.1f TF" % (t or "auto", us, flop / us / 1e6))
        print("%-24s %5.2f GF x%-2d | %s" % (name, flop / 1e9, cnt, " | ".join(res)), flush=True)
    print("per-step total (us): " + " | ".join("%s: %.0f" % (c or "auto", t) for c, t in zip(cfgs, total)))

if __name__ == "__main__":
    main()
