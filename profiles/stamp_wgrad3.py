#!/usr/bin/env python3
"""In-kernel phase stamps (s_memtime of thread 0, median over workgroups) of conv_wgrad3_kernel (narrow layers)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels"); L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
# B, H, W, Cin, Cout, k, stride, same
CASES = {"stem": (32, 32, 128, 3, 32, 7, 1, True), "tail": (32, 32, 128, 32, 3, 7, 1, True), "d1": (32, 32, 128, 6, 64, 4, 2, True),
         "out": (64, 4, 16, 512, 1, 4, 1, False)}
for name in (sys.argv[1:] or CASES):
    B, H, W, Cin, Cout, k, s, same = CASES[name]
    d = K.conv_desc(B, H, W, Cin, Cout, k, k, s, same, 1)
    x = torch.randn(B, H, W, Cin, device=dev)
    dy = torch.randn(B, d.Ho, d.Wo, Cout, device=dev)
    if Cout % 8 == 0:
        dy = dy.to(torch.bfloat16)
    job = K.wgrad_job(x, dy, k, k, torch.zeros(k, k, Cin, Cout, device=dev), torch.zeros(Cout, device=dev), stride=s, same=same, compute=K.BF16)
    for _ in range(3):
        K.conv2d_wgrad_multi([job])
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    L.load().hdrsky_debug_wgrad2_stamps(buf.data_ptr())
    K.conv2d_wgrad_multi([job])
    torch.cuda.synchronize()
    L.load().hdrsky_debug_wgrad2_stamps(None)
    t = buf.view(-1, 8).cpu()
    t = t[t[:, 0] > 0]
    med = lambda v: float(v.double().median())
    print("%-5s workgroups %d | prologue %.0f  loop %.0f (load issue %.0f, store %.0f, barrier %.0f, compute %.0f)  epilogue %.0f cycles | "
          "first start -> last end %.0f | start spread %.0f" %
          (name, t.shape[0], med(t[:, 1] - t[:, 0]), med(t[:, 2] - t[:, 1]), med(t[:, 3]), med(t[:, 4]), med(t[:, 5]), med(t[:, 6]),
           med(t[:, 7] - t[:, 2]), float(t[:, 7].max() - t[:, 0].min()), float(t[:, 0].max() - t[:, 0].min())))
