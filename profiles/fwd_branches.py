"""The two branches of the generator forward pass timed WITHOUT a profiler (HIP events on the branch streams inside
engine.ForwardGraphs replays): sun branch, encoder branch, tail, whole pass; and the same pass as ONE hipGraph with the fork /
join inside (round 4's form) for comparison.      usage: python profiles/fwd_branches.py [--steps N]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
P, synth, engine, K = (importlib.import_module(bench.PKG + "." + m) for m in ("params", "synth", "engine", "kernels"))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda", 0)
gen, sun = P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1)
ldr = torch.from_numpy(synth.make_batch(32, seed=1234)["ldr"]).to(dev)
nets = engine.Nets(gen, sun, device=dev, precise=False)
fg = engine.ForwardGraphs(nets, ldr, compute=K.BF16)
E = lambda: torch.cuda.Event(enable_timing=True)


def one(timed):
    cur = torch.cuda.current_stream()
    fg.side.wait_stream(cur); fg.main.wait_stream(cur)
    ev = {}
    with torch.cuda.stream(fg.side):
        if timed: ev["s0"] = E(); ev["s0"].record()
        fg.g_sun.replay()
        if timed: ev["s1"] = E(); ev["s1"].record()
        fg.joined.record(fg.side)
    with torch.cuda.stream(fg.main):
        if timed: ev["m0"] = E(); ev["m0"].record()
        fg.g_main.replay()
        if timed: ev["m1"] = E(); ev["m1"].record()
        fg.main.wait_event(fg.joined)
        fg.g_tail.replay()
        if timed: ev["t1"] = E(); ev["t1"].record()
    cur.wait_stream(fg.main)
    return ev


for _ in range(10):
    one(False)
torch.cuda.synchronize()
a, b = E(), E()
a.record()
for i in range(args.steps):
    ev = one(i == args.steps - 1)
b.record()
torch.cuda.synchronize()
print("ForwardGraphs (three graphs, two streams): %.4f ms per pass (mean of %d back-to-back)" % (a.elapsed_time(b) / args.steps, args.steps))
t0 = ev["s0"] if ev["s0"].elapsed_time(ev["m0"]) >= 0 else ev["m0"]
for name, e0, e1 in (("sun branch", "s0", "s1"), ("encoder branch", "m0", "m1"), ("tail (after the join)", "m1", "t1")):
    print("   %-22s %7.1f .. %7.1f us  (%6.1f us)" % (name, t0.elapsed_time(ev[e0]) * 1e3, t0.elapsed_time(ev[e1]) * 1e3, ev[e0].elapsed_time(ev[e1]) * 1e3))
# each branch alone
for name, g, s in (("sun branch alone", fg.g_sun, fg.side), ("encoder branch alone", fg.g_main, fg.main)):
    with torch.cuda.stream(s):
        for _ in range(5): g.replay()
        x, y = E(), E()
        x.record()
        for _ in range(args.steps): g.replay()
        y.record()
    torch.cuda.synchronize()
    print("   %-22s %7.1f us per replay" % (name, x.elapsed_time(y) / args.steps * 1e3))
# round 4's form: one graph, fork / join inside
one_step, _ = bench.capture_forward(torch, lambda: engine.generator_forward(nets, ldr, compute=K.BF16), False, False)
for _ in range(10): one_step()
torch.cuda.synchronize()
a.record()
for _ in range(args.steps): one_step()
b.record()
torch.cuda.synchronize()
print("one hipGraph with the fork / join inside: %.4f ms per pass" % (a.elapsed_time(b) / args.steps))
