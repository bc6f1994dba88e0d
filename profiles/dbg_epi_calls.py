#!/usr/bin/env python3
"""Diagnostic: every conv2d call of one eager B=2 bench-mode step - label, form, checksum of y and of the statistics - to a
pickle; run once per library build and diff (profiles/dbg_epi_direct.sh)."""
import importlib, os, pickle, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
trainer = importlib.import_module(PKG + ".trainer")
import test_train_gpu as T
dev = torch.device("cuda:0")
tr, nets, batch = T._mk(dev, 2, "BF16")
ldr, hdr, gt = (torch.from_numpy(batch[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
calls = []
orig = K.conv2d
def rec(x, pw, bias=None, **kw):
    y, st = orig(x, pw, bias, **kw)
    torch.cuda.synchronize()
    ys = y if isinstance(y, torch.Tensor) else y[0]
    form = "x%s %s k%d %d->%d s%d xf%d st%d ob%d res%d pair%d" % ("bf16" if x.dtype == torch.bfloat16 else "f32", tuple(x.shape), pw.KH, pw.Cin, pw.Cout,
            kw.get("stride", 1), 0 if kw.get("xf") is None else kw["xf"].mode, int(bool(kw.get("want_stats"))), int(bool(kw.get("out_bf16"))),
            int(kw.get("residual") is not None), int(kw.get("pair") is not None))
    calls.append((K._LABEL[0], form, ys.float().cpu().numpy().copy(), None if st is None else st.part.float().cpu().numpy().copy(),
                  x.float().cpu().numpy().copy()))
    return y, st
K.conv2d = rec; trainer.K.conv2d = rec
tr.step(ldr, hdr, gt, update=False)
pickle.dump(calls, open(sys.argv[1], "wb"))
print(len(calls), "calls", {k: round(v, 4) for k, v in tr.loss_dict().items()})
