#!/usr/bin/env python3
"""In-kernel phase stamps (s_memtime) of hdrsky_resconv at batch 32: median cycles per phase over the workgroups.
Phases: 0 start | 1 DMAs issued | 2 filter + plane 0 landed (barrier) | 3 main loop done | 4 K-half / strip exchange
(barrier) | 5 statistics exchange (barrier) | 6 end.  usage: python profiles/stamp_resconv.py"""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
L = importlib.import_module(PKG + "._lib")


def main():
    lib = L.load()
    lib.hdrsky_debug_resconv_stamps.argtypes = [ctypes.c_void_p]
    lib.hdrsky_debug_resconv_stamps.restype = None
    dev, B, C = torch.device("cuda:0"), 32, 128
    x = torch.randn(B, 8, 32, C, device=dev)
    w = torch.randn(3, 3, C, C, device=dev) / (9 * C) ** 0.5
    pw, pwT = K.PackedConv(w, False), K.PackedConv(w, False, transpose_flip=True)
    bias, gamma, beta = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    xb = x.to(torch.bfloat16)
    o = K.resconv_fwd(xb, pw, bias, gamma, beta, 0.1, save=True)
    nd = dict(xhat=o["xhat"], inv=o["inv"], gamma=gamma, beta=beta, slope=0.1, dgb=torch.zeros(B, 2, C, device=dev))
    cases = [("fwd act + save", lambda: K.resconv_fwd(xb, pw, bias, gamma, beta, 0.1, save=True)),
             ("bwd dgrad + norm", lambda: K.resconv_bwd(xb, pwT, norm=nd)),
             ("no-conv norm bwd", lambda: K.resconv_bwd(None, None, skip=x, norm=nd))]
    nwg = B * 8
    for name, fn in cases:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        buf = torch.zeros(nwg * 10, dtype=torch.int64, device=dev)
        lib.hdrsky_debug_resconv_stamps(buf.data_ptr())
        for _ in range(3):
            fn()          # back-to-back launches: the last one is read
        torch.cuda.synchronize()
        lib.hdrsky_debug_resconv_stamps(None)
        t = buf.cpu().numpy()
        st = t[:nwg * 8].reshape(nwg, 8).astype(np.float64)
        rt = t[nwg * 8:].reshape(nwg, 2).astype(np.float64)
        d = np.diff(st[:, :7], axis=1)
        valid = st[:, 1:7] > 0
        med = [np.median(d[valid[:, i], i]) if valid[:, i].any() else float("nan") for i in range(6)]
        tot = st[:, 6] - st[:, 0]
        clk = np.median(tot / ((rt[:, 1] - rt[:, 0]) * 10.0)) * 1e3     # s_memrealtime ticks at 100 MHz
        span = (rt[:, 1].max() - rt[:, 0].min()) * 0.01
        print("%-18s phases (cycles, median over %d WGs): %s | total %.0f | clock %.0f MHz | first start -> last end %.2f us"
              % (name, nwg, " ".join("%6.0f" % m for m in med), np.median(tot), clk, span), flush=True)


if __name__ == "__main__":
    main()
