#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): where do the GPU inference graph and the CPU oracle part ways at TRAINED-LIKE weights?
Trains bench.PARITY_FIT_STEPS steps like bench.parity_object, then compares every output tensor of the inference graph on the
first images of the held-out batch: GPU BF16X3 / BF16 against the oracle, relative max / rms error per tensor."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import step as ostep
PKG = bench.PKG
mods = {m: importlib.import_module(PKG + "." + m) for m in ("params", "synth", "engine", "trainer", "kernels", "train")}
P, synth, engine, trainer, K, train = (mods[m] for m in ("params", "synth", "engine", "trainer", "kernels", "train"))
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else bench.PARITY_FIT_STEPS
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nets = (P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1), P.init_params(P.discriminator_spec(), 2), P.init_params(P.vgg_spec(), 3))
tr = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16)
train.fit_synthetic(tr, steps, 32, seed0=0)
held = synth.make_batch_device(32, seed=999_999, device=dev)
gen_t = {k[4:]: v.detach().clone() for k, v in tr.gs.w.items() if k.startswith("gen.")}
sun_t = {k[4:]: v.detach().clone() for k, v in tr.gs.w.items() if k.startswith("sun.")}
N = engine.Nets(gen_t, sun_t, device=dev, precise=True)
ldr = held["ldr"][:n].contiguous()
o3 = engine.generator_forward(N, ldr, compute=K.BF16X3)
o16 = engine.generator_forward(N, ldr, compute=K.BF16)
torch.set_num_threads(16)
ref = ostep.inference({k: v.cpu() for k, v in gen_t.items()}, {k: v.cpu() for k, v in sun_t.items()}, ldr.cpu())
print("beta", ref["beta"].flatten().tolist(), "gamma", ref["gamma"].flatten().tolist())
for k in ("sunpose_cmf", "sun_cam1", "sun_cam2", "sun_cam3", "gamma", "beta", "sun_rad_lin", "res_out", "alpha_c3", "sky_pred_lin", "sun_pred_lin",
          "y_final_gamma", "y_final_lin"):
    r = ref[k].double()
    for name, o in (("x3", o3), ("bf16", o16)):
        g = o[k].cpu().double().reshape(r.shape)
        e = (g - r).abs()
        print("%-14s %-4s max|ref| %.4g  rel max %.3e  rel rms %.3e  worst image %d" %
              (k, name, float(r.abs().max()), float(e.max() / (r.abs().max() + 1e-30)), float((e ** 2).mean().sqrt() / ((r ** 2).mean().sqrt() + 1e-30)),
               int(e.reshape(n, -1).max(1).values.argmax())))
