#!/usr/bin/env python3
"""In-kernel phase stamps (s_memtime of thread 0, median over workgroups) of conv_wgrad2_kernel for one layer."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels"); L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
CASES = {"res": (32, 8, 32, 128, 128, 3, 1), "l2a": (32, 16, 64, 32, 64, 3, 1), "d4": (64, 4, 16, 256, 512, 4, 1), "l1b": (32, 32, 128, 32, 32, 7, 1),
         "dec2": (32, 32, 128, 64, 32, 3, 1)}
for name in (sys.argv[1:] or CASES):
    B, H, W, Cin, Cout, k, s = CASES[name]
    d = K.conv_desc(B, H, W, Cin, Cout, k, k, s, True, 1)
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, d.Ho, d.Wo, Cout, device=dev).to(torch.bfloat16)
    job = K.wgrad_job(x, dy, k, k, torch.zeros(k, k, Cin, Cout, device=dev), torch.zeros(Cout, device=dev), stride=s, compute=K.BF16)
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
    print("%-5s workgroups %d, tiles/wg %.0f | prime %.0f  loop %.0f (wait %.0f, issue %.0f, compute %.0f)  epilogue %.0f cycles | first start -> last end %.0f" %
          (name, t.shape[0], med(t[:, 6]), med(t[:, 1] - t[:, 0]), med(t[:, 2] - t[:, 1]), med(t[:, 3]), med(t[:, 4]), med(t[:, 5]),
           med(t[:, 7] - t[:, 2]), float(t[:, 7].max() - t[:, 0].min())))
