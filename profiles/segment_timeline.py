"""Per-segment timeline of the captured three-stream training step WITHOUT a profiler attached (rocprofv3's interception
makes the host too slow to keep three streams fed, which opens gaps that the plain run does not have): a pair of HIP timing
events around every segment's graph replay, recorded on the segment's stream through Trainer.replay's hooks, for the last
of a run of back-to-back steps.  Prints start / end / duration per segment (microseconds from the first start), per stream,
and how long each stream idles.      usage: python profiles/segment_timeline.py [--da PARTS] [--steps N]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
P, synth, trainer, K = (importlib.import_module(bench.PKG + "." + m) for m in ("params", "synth", "trainer", "kernels"))
ap = argparse.ArgumentParser()
ap.add_argument("--da", default="")
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
nets = [P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1), P.init_params(P.discriminator_spec(), 2),
        P.init_params(P.vgg_spec(), 3)]
b = synth.make_batch(32, seed=1234)
ldr, hdr, gt = (torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
tr = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16, distortion_aware=args.da or False)
tr.capture(ldr, hdr, gt)
for _ in range(5):
    tr.replay()
names = [(n, si) for n, si, _, fn in tr._segs if n not in tr._skip(True)]
ev0 = {n: torch.cuda.Event(enable_timing=True) for n, _ in names}
ev1 = {n: torch.cuda.Event(enable_timing=True) for n, _ in names}
pre = {n: (lambda n=n: ev0[n].record(torch.cuda.current_stream())) for n, _ in names}
post = {n: (lambda n=n: ev1[n].record(torch.cuda.current_stream())) for n, _ in names}
torch.cuda.synchronize()
t_all0, t_all1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t_all0.record()
for i in range(args.steps):
    last = i == args.steps - 1
    tr.replay(hooks=post if last else None, pre_hooks=pre if last else None)
t_all1.record()
torch.cuda.synchronize()
print("step %.4f ms (mean of %d back-to-back)" % (t_all0.elapsed_time(t_all1) / args.steps, args.steps))
first = min(names, key=lambda x: t_all0.elapsed_time(ev0[x[0]]))[0]
rows = [(ev0[first].elapsed_time(ev0[n]) * 1e3, ev0[first].elapsed_time(ev1[n]) * 1e3, si, n) for n, si in names]
end = max(r[1] for r in rows)
for si in sorted({r[2] for r in rows}):
    mine = sorted(r for r in rows if r[2] == si)
    busy = sum(e - s for s, e, _, _ in mine)
    print("stream %d: busy %.0f of %.0f us" % (si, busy, end))
    prev = 0.0
    for s, e, _, n in mine:
        print("   %-12s %7.0f .. %7.0f  (%5.0f us)%s" % (n, s, e, e - s, "   <- idle %.0f" % (s - prev) if s - prev > 20 else ""))
        prev = e
