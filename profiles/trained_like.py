#!/usr/bin/env python3
"""How many synthetic training steps until the generator's output is 'trained-like' (PSNR(y_gamma, target) >= 30 dB)?
Trains with the product's own captured step (bench mode, HDRSKY_BF16) on seeded device-side synthetic batches
(<pkg>/train.py::fit_synthetic - the loop `python -m <pkg>.train` runs) and evaluates, every --every steps, the inference
graph on a held-out batch in both compute modes: PSNR against the log-compressed target and between the modes.
  python profiles/trained_like.py [--steps 3000] [--every 250] [--lr 1e-4]
"""
import argparse, importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000); ap.add_argument("--every", type=int, default=250)
ap.add_argument("--lr", type=float, default=1e-4); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--no-jpeg", action="store_true")
a = ap.parse_args()
P, synth, trainer, K, engine, train = (importlib.import_module(PKG + "." + m) for m in ("params", "synth", "trainer", "kernels", "engine", "train"))
dev = torch.device("cuda:0")
nets = [P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1), P.init_params(P.discriminator_spec(), 2), P.init_params(P.vgg_spec(), 3)]
tr = trainer.Trainer(*nets, device=dev, lr=a.lr, compute=K.BF16)
held = synth.make_batch_device(a.batch, seed=999_999, device=dev, jpeg=not a.no_jpeg)
done, t0 = 0, time.perf_counter()
while done <= a.steps:
    print(done, train.quality_report(tr, held["ldr"], held["hdr_t"]), "%.1f s" % (time.perf_counter() - t0), flush=True)
    if done == a.steps:
        break
    train.fit_synthetic(tr, a.every, a.batch, seed0=done, jpeg=not a.no_jpeg)
    done += a.every
