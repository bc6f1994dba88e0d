#!/usr/bin/env python3
"""How many dependent launches per second do 1 / 2 / 3 / 4 HIP streams sustain when every stream replays a hipGraph of N tiny
kernels (hdrsky_zero of 1 KB: ~2 us of work)?  If k streams sustain k times the rate of one, the command processor is not
what the three-stream training step (284 launches in 2.53 ms = 112 k launches/s) waits for.
Second part: the same with a mid-size kernel in the chain (a 3x3 64->64 conv at 32x128, B=16) on stream 0 and tiny kernels on the
others - does a stream of tiny kernels slow down beside real work?"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
dev = torch.device("cuda:0")
N = 200
streams = [torch.cuda.Stream() for _ in range(4)]
bufs = [torch.empty(256, device=dev) for _ in range(4)]


def graph_of(fn, s):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(N):
            fn()
    return g


tiny = [graph_of(lambda b=b: K.zero_(b), s) for b, s in zip(bufs, streams)]
x = torch.randn(16, 32, 128, 64, device=dev).to(torch.bfloat16)
pw = K.PackedConv(torch.randn(3, 3, 64, 64, device=dev) * 0.05, precise=False)
bias = torch.zeros(64, device=dev)
conv = [graph_of(lambda: K.conv2d(x, pw, bias, compute=K.BF16, out_slope=0.0, out_bf16=True), s) for s in streams[:3]]


def run(graphs, reps=5):
    torch.cuda.synchronize()
    for _ in range(2):
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


for k in (1, 2, 3, 4):
    us = run([(tiny[i], streams[i]) for i in range(k)])
    print("%d stream(s) x %d tiny launches: %8.1f us per round = %6.2f us per launch per stream, %7.0f k launches/s in total" % (
        k, N, us, us / N, k * N / us * 1e3), flush=True)
for k in (1, 2, 3):
    us = run([(conv[i], streams[i]) for i in range(k)])
    print("%d stream(s) x %d conv launches (3x3 64->64 @32x128 B=16, 4.8 GFLOP): %8.1f us per round = %6.2f us per launch per stream, %6.1f TFLOP/s" % (
        k, N, us, us / N, k * N * 4.83e9 / us / 1e6), flush=True)
us = run([(conv[0], streams[0]), (tiny[1], streams[1]), (tiny[2], streams[2])])
print("conv chain on stream 0 beside two tiny chains: %8.1f us per round (%6.2f us per launch of the slowest chain)" % (us, us / N), flush=True)
