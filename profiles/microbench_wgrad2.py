#!/usr/bin/env python3
"""Weight-gradient launches of the training step's layers (batch 32, bf16 operands), register-staged kernel
(HDRSKY_WGRAD2=0) against the LDS-DMA ring kernel (default): main + reduce launch per call, timed back to back from one
hipGraph with HIP events.  --group: the layers of each trainer segment in ONE call, as the step issues them."""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
ap = argparse.ArgumentParser(); ap.add_argument("--group", action="store_true"); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="", help="substring filter on the layer / group name")
args = ap.parse_args()
dev = torch.device("cuda:0")
# (name, B, H, W, Cin, Cout, k, stride)
LAYERS = [("res 3x3 128->128 @8x32", 32, 8, 32, 128, 128, 3, 1), ("l3b 3x3 128->128 @8x32", 32, 8, 32, 128, 128, 3, 1),
          ("l3a 3x3 64->128 @8x32", 32, 8, 32, 64, 128, 3, 1), ("l2b 3x3 64->64 @16x64", 32, 16, 64, 64, 64, 3, 1),
          ("l2a 3x3 32->64 @16x64", 32, 16, 64, 32, 64, 3, 1), ("l1b 7x7 32->32 @32x128", 32, 32, 128, 32, 32, 7, 1),
          ("dec3 3x3 128->64 @16x64", 32, 16, 64, 128, 64, 3, 1), ("dec2 3x3 64->32 @32x128", 32, 32, 128, 64, 32, 3, 1),
          ("conv2_d 3x3 s2 32->64 @32x128", 32, 32, 128, 32, 64, 3, 2), ("conv3_d 3x3 s2 64->128 @16x64", 32, 16, 64, 64, 128, 3, 2),
          ("sunrad d2 4x4 s2 64->128 @16x64", 32, 16, 64, 64, 128, 4, 2), ("sunrad d3 4x4 s2 128->256 @8x32", 32, 8, 32, 128, 256, 4, 2),
          ("sunrad d4 4x4 256->512 @4x16", 32, 4, 16, 256, 512, 4, 1), ("disc d2 4x4 s2 64->128 @16x64 B=64", 64, 16, 64, 64, 128, 4, 2),
          ("disc d3 4x4 s2 128->256 @8x32 B=64", 64, 8, 32, 128, 256, 4, 2), ("disc d4 4x4 256->512 @4x16 B=64", 64, 4, 16, 256, 512, 4, 1)]
LAYERS += [("hires d2 4x4 s2 64->128 @64x256 B=8", 8, 64, 256, 64, 128, 4, 2), ("hires d2 4x4 s2 64->128 @64x256 B=16", 16, 64, 256, 64, 128, 4, 2),
           ("hires conv2_d 3x3 s2 32->64 @128x512 B=8", 8, 128, 512, 32, 64, 3, 2), ("hires conv3_d 3x3 s2 64->128 @64x256 B=8", 8, 64, 256, 64, 128, 3, 2),
           ("hires d3 4x4 s2 128->256 @32x128 B=8", 8, 32, 128, 128, 256, 4, 2), ("hires res 3x3 128->128 @32x128 B=8", 8, 32, 128, 128, 128, 3, 1)]
# narrow layers (<= 8 input channels or < 32 output channels: register-staged kernel only; x fp32 as in the step)
LAYERS += [("narrow 7x7 3->32 @32x128", 32, 32, 128, 3, 32, 7, 1), ("narrow 7x7 32->3 @32x128", 32, 32, 128, 32, 3, 7, 1),
           ("narrow sunrad d1 4x4 s2 6->64 @32x128", 32, 32, 128, 6, 64, 4, 2), ("narrow disc d1 4x4 s2 6->64 @32x128 B=64", 64, 32, 128, 6, 64, 4, 2),
           ("narrow disc out 4x4 512->1 @4x16 valid B=64", 64, 4, 16, 512, 1, 4, 1)]
GROUPS = {"sunpose": [1, 2, 3, 4, 5], "decoders x2": [6, 7, 6, 7], "encoder": [8, 9], "sunrad": [10, 11, 12], "disc": [13, 14, 15],
          "res x12": [0] * 12}


def gtime(fn, iters):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def job(l):
    name, B, H, W, Cin, Cout, k, s = l
    same = "valid" not in name
    d = K.conv_desc(B, H, W, Cin, Cout, k, k, s, same, 1)
    x = torch.randn(B, H, W, Cin, device=dev)
    if not name.startswith("narrow"):
        x = x.to(torch.bfloat16)
    dy = torch.randn(B, d.Ho, d.Wo, Cout, device=dev)
    if Cout % 8 == 0:
        dy = dy.to(torch.bfloat16)
    return K.wgrad_job(x, dy, k, k, torch.zeros(k, k, Cin, Cout, device=dev), torch.zeros(Cout, device=dev), stride=s, same=same, compute=K.BF16)


def flop(l):
    name, B, H, W, Cin, Cout, k, s = l
    if "valid" in name:
        return 2.0 * B * (H - k + 1) * (W - k + 1) * k * k * Cin * Cout
    return 2.0 * B * (H // s) * (W // s) * k * k * Cin * Cout


sets = [(n, [LAYERS[i] for i in idx]) for n, idx in GROUPS.items()] if args.group else [(l[0], [l]) for l in LAYERS]
for name, ls in sets:
    if args.only and args.only not in name:
        continue
    jobs = [job(l) for l in ls]
    fl = sum(flop(l) for l in ls)
    res = {}
    hook = "HDRSKY_WGRAD3" if name.startswith("narrow") else "HDRSKY_WGRAD2"   # v1 = the register-staged kernel in both cases
    for v in ("0", "1"):
        os.environ[hook] = v
        res[v] = gtime(lambda: K.conv2d_wgrad_multi(jobs), args.iters)
    print("%-40s %7.2f GFLOP  v1 %7.1f us (%6.1f TF/s)   v2 %7.1f us (%6.1f TF/s)   x%.2f" %
          (name, fl / 1e9, res["0"], fl / res["0"] / 1e6, res["1"], fl / res["1"] / 1e6, res["0"] / res["1"]), flush=True)
    os.environ.pop(hook)
