import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
PKG = bench.PKG
mods = {m: importlib.import_module(PKG + "." + m) for m in ("params", "synth", "engine", "trainer", "kernels", "train")}
P = mods["params"]
dev = torch.device("cuda:0")
nets = (P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1), P.init_params(P.discriminator_spec(), 2), P.init_params(P.vgg_spec(), 3))
for mode in ("x3", "bf16"):
    os.environ["HDRSKY_PARITY_FIT"] = mode
    p = bench.parity_object(torch, mods, dev, nets, 32, bench.oracle_outputs_fn(torch, 8))
    print(mode, p, flush=True)
