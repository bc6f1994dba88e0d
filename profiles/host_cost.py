"""Host-side cost of one replayed training step (time to enqueue, no sync) vs its GPU time."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hdrsky_amd as hs
params, synth, trainer, K = (importlib.import_module(hs.__name__ + "." + m) for m in ("params", "synth", "trainer", "kernels"))
dev = torch.device("cuda:0")
gen = params.init_params(params.generator_spec(), 0); sun = params.init_params(params.sunpose_spec(), 1)
dis = params.init_params(params.discriminator_spec(), 2); vgg = params.init_params(params.vgg_spec(), 3)
b = synth.make_batch(32, seed=1234)
ldr, hdr, gt = (torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16)
tr.capture(ldr, hdr, gt)
for _ in range(5): tr.replay()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): tr.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per step %.3f ms; wall per step %.3f ms" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
# per-graph launch cost
import collections
cost = collections.OrderedDict()
for name, gr in tr._graphs.items():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): gr.replay()
    cost[name] = (time.perf_counter() - t0) / 20 * 1e6
    torch.cuda.synchronize()
print({k: round(v) for k, v in cost.items()}, "sum", round(sum(cost.values())))
# ---- segment timeline from HIP events (no profiler in the way) ----------------------------------------------------
names = [n for n, *_ in tr._segs]
st = {n: torch.cuda.Event(enable_timing=True) for n in names}
en = {n: torch.cuda.Event(enable_timing=True) for n in names}
base = torch.cuda.Event(enable_timing=True)
for _ in range(3): tr.replay()
torch.cuda.synchronize()
base.record()
tr.replay(pre_hooks={n: (lambda n=n: st[n].record()) for n in names}, hooks={n: (lambda n=n: en[n].record()) for n in names})
torch.cuda.synchronize()
print("segment timeline (us from step start): name stream start end dur")
for n, si, d, f in tr._segs:
    a, b_ = base.elapsed_time(st[n]) * 1e3, base.elapsed_time(en[n]) * 1e3
    print("  %-12s s%d %7.0f %7.0f %6.0f   deps %s" % (n, si, a, b_, b_ - a, list(d)))
# ---- each segment's graph alone (nothing else on the GPU) ---------------------------------------------------------
print("segment alone (us):")
tot = {}
for n, si, d, f in tr._segs:
    if f is None: continue
    gr = tr._graphs[n]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(tr._streams[si]):
        gr.replay(); torch.cuda.synchronize()
        e0.record(); gr.replay(); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    tot[si] = tot.get(si, 0) + us
    print("  %-12s s%d %7.0f" % (n, si, us))
print("sum per stream:", {k: round(v) for k, v in tot.items()}, "all:", round(sum(tot.values())))
