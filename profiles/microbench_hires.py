#!/usr/bin/env python3
"""BASELINE configs[4] datapoints (128x512 panoramas, 8 per GPU): generator encoder + both decoders, the res stack with
plain vs distortion-aware 3x3 convolutions, the discriminator and VGG16 forward.  (The faithful 128x512 sun-pose net has
12.9 G parameters - SURVEY.md section 8d - and is not part of this measurement.)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdrsky_amd as hs
params, engine, K, ops_da = (importlib.import_module(hs.__name__ + "." + m) for m in ("params", "engine", "kernels", "distortion_aware_ops"))
disc_mod, vgg_mod = (importlib.import_module(hs.__name__ + "." + m) for m in ("discriminator", "vgg16"))
dev = torch.device("cuda:0")
B, H, W = 8, 128, 512


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


gen = params.init_params(params.generator_spec(H, W), 0)
nets = engine.Nets(gen, None, device=dev, precise=False, im_height=H, im_width=W)
ldr = torch.rand(B, H, W, 3, device=dev)
rad = torch.rand(B, H, W, 3, device=dev)
res = engine.encode(nets, ldr, K.BF16)
ms = timeit(lambda: engine.encode(nets, ldr, K.BF16))
print("encoder (3 convs + 6 res blocks)      %.3f ms  (%.1f TFLOP/s of 16.3 GFLOP/img)" % (ms, B * 16.32e9 / ms / 1e9))
ms_da = timeit(lambda: engine.encode(nets, ldr, K.BF16, distortion_aware=True))
print("encoder, distortion-aware res blocks  %.3f ms  (the variant generator.py:14,18 keeps commented out)" % ms_da)
ms = timeit(lambda: (engine.decode(nets, res, "f", ldr, K.BF16), engine.decode(nets, res, "u", rad, K.BF16)))
print("sky + sun decoders                    %.3f ms  (%.1f TFLOP/s of 10.9 GFLOP/img)" % (ms, B * 10.9e9 / ms / 1e9))
x = torch.randn(B, H // 4, W // 4, 128, device=dev)
w = torch.randn(3, 3, 128, 128, device=dev) / 34.0
pw = K.PackedConv(w, False); bias = torch.zeros(128, device=dev)
ms_p = timeit(lambda: K.conv2d(x, pw, bias, want_stats=True), 50)
da = ops_da.conv2d(128, 3, compute=K.BF16); da(x)
ms_d = timeit(lambda: da(x), 50)
flop = 2.0 * B * (H // 4) * (W // 4) * 1152 * 128
print("res conv 3x3 128->128 @32x128: plain %.1f us (%.0f TF), distortion-aware %.1f us (%.0f TF)" %
      (ms_p * 1e3, flop / ms_p / 1e9, ms_d * 1e3, flop / ms_d / 1e9))
dis = disc_mod.model(device=dev, compute=K.BF16)
hdr = torch.rand(B, H, W, 3, device=dev)
ms = timeit(lambda: dis([ldr, hdr], training=False))
print("discriminator forward (eval)          %.3f ms  (%.1f TFLOP/s of 6.66 GFLOP/img)" % (ms, B * 6.657e9 / ms / 1e9))
vgg = vgg_mod.Vgg16(weights=params.init_params(params.vgg_spec(), 3), device=dev, compute=K.BF16)
ms = timeit(lambda: vgg(hdr))
print("VGG16 -> pool3 forward                %.3f ms  (%.1f TFLOP/s of 24.4 GFLOP/img)" % (ms, B * 24.39e9 / ms / 1e9))
