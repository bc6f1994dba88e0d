#!/usr/bin/env python3
"""K eager updating steps of the 32x128 training workload (bench.py's `train` set-up: same synthetic batch, same Trainer)
and nothing else - the subject of rocprofv3 --pmc passes, which do not see the kernels of a replayed hipGraph.
run_steps.py [K=6]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
PKG = bench.PKG
m = {k: importlib.import_module(PKG + "." + k) for k in ("params", "synth", "trainer", "kernels")}
params, synth, trainer, K = m["params"], m["synth"], m["trainer"], m["kernels"]
dev = torch.device("cuda", 0)
b = synth.make_batch(32, seed=1234)
ldr, hdr, gt = (torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
tr = trainer.Trainer(params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
                     params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3),
                     device=dev, precise=False, compute=K.BF16)
tr.repack()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    out = tr.step(ldr, hdr, gt, update=True)
torch.cuda.synchronize()
assert torch.isfinite(out["y_final_lin"]).all()
print("ok")
