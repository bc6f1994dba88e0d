"""Marginal cost of every C-ABI entry point inside the captured training step.

Kernel durations from a profile of the three-stream step overlap each other and do not add up to the step time.  This
script re-captures the step once per entry point with that entry point's launches REMOVED (the ctypes attribute is
replaced by a no-op on the loaded library object - results are garbage, timing is what is measured) and prints how much
shorter the step gets: the step's sensitivity to that family of launches.  Profiling aid only; nothing in the package
knows about it.      usage: python profiles/ablate_step.py [symbol ...]
"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
P, synth, trainer, K, L = (importlib.import_module(bench.PKG + "." + m) for m in ("params", "synth", "trainer", "kernels", "_lib"))
dev = torch.device("cuda", 0)
nets = [P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1), P.init_params(P.discriminator_spec(), 2),
        P.init_params(P.vgg_spec(), 3)]
b = synth.make_batch(32, seed=1234)
ldr, hdr, gt = (torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
lib = L.load()


def run(steps=40):
    tr = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16)
    tr.capture(ldr, hdr, gt)
    for _ in range(5):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    del tr
    torch.cuda.empty_cache()
    return dt


base = [run() for _ in range(2)]
print("baseline %.4f %.4f ms" % tuple(base), flush=True)
base = min(base)
# Default: entry points whose outputs nothing downstream uses as an index.  Removing a producer of index data (pooling
# routes, arg-max bins, gather tables) makes its consumers read out of bounds - a GPU memory fault: never ablate those.
SAFE = ["hdrsky_conv2d_wgrad_multi_det", "hdrsky_norm_act_bwd", "hdrsky_rmsprop_fc_fused", "hdrsky_rmsprop",
        "hdrsky_fc_dgrad", "hdrsky_dgb_reduce", "hdrsky_bn_act_bwd", "hdrsky_affine_act_bwd", "hdrsky_conv_pack_weights_multi"]
names = sys.argv[1:] or SAFE
rows = []
for n in names:
    real = getattr(lib, n)
    setattr(lib, n, lambda *a, **k: 0)
    try:
        dt = run(25)
    except Exception as e:      # an entry point whose result the host needs
        dt = None
    setattr(lib, n, real)
    if dt is not None:
        rows.append((base - dt, n))
        print("%-40s %8.4f ms  (%+.3f)" % (n, dt, dt - base), flush=True)
rows.sort(reverse=True)
print("---- step time saved when the launches of an entry point are removed (ms of %.3f)" % base)
for d, n in rows[:25]:
    print("%-40s %7.3f" % (n, d))
