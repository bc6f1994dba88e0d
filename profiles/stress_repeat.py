"""One-off stress: replay the captured batch-32 step many times from the same state; everything that involves no
atomics must be bit-identical every time, the rest within rounding."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdrsky_amd as hs
params, synth, trainer, K = (importlib.import_module(hs.__name__ + "." + m) for m in ("params", "synth", "trainer", "kernels"))
dev = torch.device("cuda:0")
gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
bt = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_batch(32, seed=7).items()}
tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16,
                     distortion_aware=os.environ.get("STRESS_DA", "") or False)      # STRESS_DA=all: the distortion-aware variant
w0g, w0d = tr.gs.flat.clone(), tr.ds.flat.clone()
tr.capture(bt["ldr"], bt["hdr_t"], bt["sunpose_gt"])
ref = None
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for it in range(N):
    tr.gs.flat.copy_(w0g); tr.ds.flat.copy_(w0d)
    tr.repack()
    tr.replay(update=False)
    torch.cuda.synchronize()
    T = tr._T
    snap = dict(y=T["y_lin"].clone(), fc1=tr.gs.g["sun.fc1.kernel"].clone(), fc2=tr.gs.g["sun.fc2.kernel"].clone(),
                dP3=T["dP3"].clone(), dyg=T["dyg"].clone(), dadv=T["d_adv"].clone(), cam1=T["cams"][0].clone())
    gn = (float(tr.gs.grad.double().norm()), float(tr.ds.grad.double().norm()))
    if ref is None:
        ref, gref = snap, gn
        continue
    for k in snap:
        if not torch.equal(snap[k], ref[k]):
            bad += 1
            print("iteration", it, k, "differs: max abs", float((snap[k] - ref[k]).abs().max()), "of", float(ref[k].abs().max()), flush=True)
    for a, b_, n in zip(gn, gref, ("gen+sun grad norm", "disc grad norm")):
        if abs(a - b_) > 2e-4 * b_:
            bad += 1
            print("iteration", it, n, a, "vs", b_, flush=True)
print("stress: %d iterations, %d discrepancies" % (N, bad))
