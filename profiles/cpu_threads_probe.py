import sys, time, importlib, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
P = importlib.import_module(PKG + ".params"); synth = importlib.import_module(PKG + ".synth")
from oracle import step as ostep
nets = [P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1), P.init_params(P.discriminator_spec(), 2), P.init_params(P.vgg_spec(), 3)]
tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
gen, sun, dis, vgg = (tt(d) for d in nets)
b = synth.make_batch(32, seed=1234)
ldr, hdr, gt = (torch.from_numpy(b[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "default threads", torch.get_num_threads())
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    for n in (4, 32):
        ostep.train_step_grads(gen, sun, dis, vgg, ldr[:n], hdr[:n], gt[:n])
        t0 = time.perf_counter(); ostep.train_step_grads(gen, sun, dis, vgg, ldr[:n], hdr[:n], gt[:n]); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); ostep.inference(gen, sun, ldr[:n]); df = time.perf_counter() - t0
        print("threads %3d batch %2d: train %.2f s (%.1f img/s)  fwd %.2f s (%.1f img/s)" % (th, n, dt, n / dt, df, n / df), flush=True)
