"""Compute-unit masks on the two branches of the generator forward pass (engine.ForwardGraphs on streams from
kernels.masked_stream): does reserving compute units for the sun branch - a dependent chain of ~35 mostly small launches whose
every launch otherwise waits for a slot behind the encoder branch's 8-10 us conv workgroups - shorten the pass?
usage: python profiles/cu_mask_fwd.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
P, synth, engine, K = (importlib.import_module(bench.PKG + "." + m) for m in ("params", "synth", "engine", "kernels"))
dev = torch.device("cuda", 0)
gen, sun = P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1)
ldr = torch.from_numpy(synth.make_batch(32, seed=1234)["ldr"]).to(dev)
nets = engine.Nets(gen, sun, device=dev, precise=False)
E = lambda: torch.cuda.Event(enable_timing=True)


def time_fg(fg, steps=100):
    for _ in range(10):
        fg.replay()
    torch.cuda.synchronize()
    a, b = E(), E()
    a.record()
    for _ in range(steps):
        fg.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


one_step, _ = bench.capture_forward(torch, lambda: engine.generator_forward(nets, ldr, compute=K.BF16), False, False)
for _ in range(10): one_step()
torch.cuda.synchronize()
a, b = E(), E(); a.record()
for _ in range(100): one_step()
b.record(); torch.cuda.synchronize()
print("one hipGraph, fork / join inside:                 %.4f ms" % (a.elapsed_time(b) / 100), flush=True)
print("two graphs, two unmasked streams:                 %.4f ms" % time_fg(engine.ForwardGraphs(nets, ldr, compute=K.BF16)), flush=True)
for name, sun_cu, enc_cu in (("sun all | encoder 64..255", None, (64, 256)), ("sun all | encoder 128..255", None, (128, 256)),
                             ("sun all | encoder 32..255", None, (32, 256)),
                             ("sun 0..63 | encoder 64..255", (0, 64), (64, 256)), ("sun 0..127 | encoder 128..255", (0, 128), (128, 256)),
                             ("sun 0..31 | encoder 32..255", (0, 32), (32, 256)), ("sun 0..191 | encoder 64..255", (0, 192), (64, 256))):
    side = K.masked_stream(*sun_cu) if sun_cu else None
    main = K.masked_stream(*enc_cu)
    fg = engine.ForwardGraphs(nets, ldr, compute=K.BF16, main_stream=main, side_stream=side)
    print("two graphs, %-37s %.4f ms" % (name + ":", time_fg(fg)), flush=True)
    del fg
