#!/usr/bin/env python3
"""Saturated throughput of conv_igemm_kernel tile shapes: three HIP streams each replay a hipGraph of N identical conv launches
(what a launch costs when the chip is full of waves of its own kind - the regime of the three-stream training step - as opposed
to its latency alone on the chip, which is what round 1-3's tile table was tuned on), per layer shape and tile (HDRSKY_TILE hook).
usage: python profiles/conv_throughput.py"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HDRSKY_EXPERIMENTS"] = "1"
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
HK = importlib.import_module(PKG + ".hooks")
L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
N = 60
streams = [torch.cuda.Stream() for _ in range(3)]


def run(graphs, reps=4):
    torch.cuda.synchronize()
    for g, s in graphs:
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, s in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


SHAPES = [("vgg 64->64 @32x128 B16", 16, 32, 128, 64, 64, 3, True), ("vgg 128->128 @16x64 B16", 16, 16, 64, 128, 128, 3, True),
          ("vgg 256->256 @8x32 B16", 16, 8, 32, 256, 256, 3, True), ("dec 64->32 @32x128 B32 (f32 out+stats)", 32, 32, 128, 64, 32, 3, False),
          ("dec 128->64 @16x64 B32", 32, 16, 64, 128, 64, 3, False), ("sun 32->32 7x7 @32x128 B32", 32, 32, 128, 32, 32, 7, False)]
TILES = [None, "2,4,4,1,32,1", "2,2,4,2,32,1", "2,4,4,2,32,1", "1,4,4,1,32,1", "4,2,4,1,32,1", "1,8,4,1,32,1", "4,1,4,4,32,1", "2,2,4,4,32,1", "4,1,2,4,32,1", "1,4,4,4,32,1"]
for name, B, H, W, C, F, k, bf in SHAPES:
    x = torch.randn(B, H, W, C, device=dev)
    if bf:
        x = x.to(torch.bfloat16)
    pw = K.PackedConv(torch.randn(k, k, C, F, device=dev) * 0.05, precise=False)
    bias = torch.zeros(F, device=dev)
    flop = 2.0 * B * H * W * k * k * C * F
    for tile in TILES:
        if tile is None:
            os.environ.pop("HDRSKY_TILE", None)
        else:
            os.environ["HDRSKY_TILE"] = tile
        HK.reload()
        fn = (lambda: K.conv2d(x, pw, bias, compute=K.BF16, out_slope=0.0, out_bf16=True)) if bf else \
             (lambda: K.conv2d(x, pw, bias, compute=K.BF16, want_stats=True))
        try:
            fn(); torch.cuda.synchronize()
        except Exception as e:
            print("%-40s tile %-14s unsupported (%s)" % (name, tile, str(e)[:40]), flush=True)
            continue
        gs = []
        for s in streams:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(N):
                    fn()
            gs.append((g, s))
        one = run(gs[:1]) / N
        three = run(gs) / N
        d = K.conv_desc(B, H, W, C, F, k, k); d.compute = K.BF16
        print("%-40s tile %-14s alone %6.2f us (%5.0f TF)   3 streams %6.2f us per launch -> %5.0f TFLOP/s aggregate   [%s]" % (
            name, tile or "(table)", one, flop / one / 1e6, three, 3 * flop / three / 1e6, K.conv_kernel_name(d)[18:40]), flush=True)
os.environ.pop("HDRSKY_TILE", None); HK.reload()
