"""The fused Dense update (fc_xtdy_kernel<fused>) under the non-temporal policies of HDRSKY_FC_NT (bit 0: image stores, 1: w / ms
stores, 2: w / ms loads), alone on the chip, fc1's 8192x4096 kernel at M = 32.   usage: python profiles/microbench_fc_nt.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HDRSKY_EXPERIMENTS"] = "1"
import bench
K = importlib.import_module(bench.PKG + ".kernels"); HK = importlib.import_module(bench.PKG + ".hooks")
dev = torch.device("cuda", 0)
Kd, N, M = 8192, 4096, 32
w = torch.randn(Kd, N, device=dev) * 0.01
ms = torch.zeros(Kd, N, device=dev)
pf = K.PackedFC(w, precise=False)
db = torch.zeros(N, device=dev)
x, dy = torch.randn(M, Kd, device=dev), torch.randn(M, N, device=dev) * 1e-3
for rep in range(2):
    for nt in range(8):
        os.environ["HDRSKY_FC_NT"] = str(nt); HK.reload()
        us = bench._graph_time(torch, lambda: K.rmsprop_fc_fused(w, ms, x, dy, pf, 1e-4, db=db), 20, warm=3)
        print("HDRSKY_FC_NT=%d  %7.1f us  %5.0f GB/s (%.3f of 8 TB/s)" % (nt, us, 20 * Kd * N / us / 1e3, 20 * Kd * N / us / 1e3 / 8000), flush=True)
