#!/usr/bin/env python3
"""Phase breakdown of one conv launch from in-kernel s_memtime stamps (diagnostic path only).
Shares (cycles per workgroup): prologue | A staging | main loop (B ring + MFMA) | epilogue.
usage: python profiles/stamp_conv.py <layer-substring> ["tile;tile;..."]"""
import ctypes
import importlib
import os
import sys

import numpy as np
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
sys.path.insert(0, os.path.join(ROOT, "profiles"))
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
L = importlib.import_module(PKG + "._lib")
from microbench_conv import LAYERS  # noqa: E402


def main():
    lib = L.load()
    lib.hdrsky_debug_conv_stamps.argtypes = [ctypes.c_void_p]
    lib.hdrsky_debug_conv_stamps.restype = None
    dev = torch.device("cuda:0")
    B = 32
    only = sys.argv[1] if len(sys.argv) > 1 else "res 128"
    tiles = sys.argv[2].split(";") if len(sys.argv) > 2 else [""]
    for (name, H, W, Cin, Cout, k, stride, _cnt) in LAYERS:
        if only not in name:
            continue
        up, stats = 1, True
        x = torch.randn(B, H, W, Cin, device=dev)
        kw = dict(want_stats=stats)
        if os.environ.get("STAMP_BF16", "0") == "1":      # the VGG16 chain's form: bf16 in, bf16 out, ReLU epilogue, no statistics
            x = x.to(torch.bfloat16); kw = dict(out_bf16=True, out_slope=0.0)
        w = torch.randn(k, k, Cin, Cout, device=dev) / (k * k * Cin) ** 0.5
        pw = K.PackedConv(w, precise=False)
        bias = torch.zeros(Cout, device=dev)
        for t in tiles:
            if t:
                _hook("HDRSKY_TILE", t)
            else:
                _hook("HDRSKY_TILE", None)
            buf = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
            for _ in range(20):
                K.conv2d(x, pw, bias, stride=stride, upsample=up, **kw)
            torch.cuda.synchronize()
            lib.hdrsky_debug_conv_stamps(buf.data_ptr())
            K.conv2d(x, pw, bias, stride=stride, upsample=up, **kw)
            torch.cuda.synchronize()
            lib.hdrsky_debug_conv_stamps(None)
            raw = buf.cpu().numpy()
            s = raw.reshape(-1, 8)
            s = s[s[:, 0] != 0]
            d = np.stack([s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 5] - s[:, 3]], 1).astype(np.float64)
            tot = (s[:, 5] - s[:, 0]).astype(np.float64)
            rt = (s[:, 7] - s[:, 6]).astype(np.float64)
            mhz = np.median(tot / np.maximum(rt, 1)) * 100.0
            span = (s[:, 7].max() - s[:, 6].min()) / 100.0
            print("%-24s tile %-12s WGs %5d | cycles/WG: prologue %6.0f  stageA %6.0f  mainloop %6.0f  epilogue %6.0f | "
                  "total %6.0f cyc = %.2f us @ %.0f MHz | kernel span %.2f us"
                  % ((name, t or "auto", len(s)) + tuple(np.median(d, 0)) + (np.median(tot), np.median(tot) / mhz, mhz, span)),
                  flush=True)


if __name__ == "__main__":
    main()
