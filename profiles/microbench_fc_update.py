"""Standalone timing of the Dense update / weight-gradient launches (fc_update.hip) - usage: python profiles/microbench_fc_update.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
K = importlib.import_module(bench.PKG + ".kernels")
dev = torch.device("cuda", 0)
for Kd, N in ((8192, 4096), (4096, 2048)):
    w = torch.randn(Kd, N, device=dev) * 0.01
    g, ms = torch.randn(Kd, N, device=dev) * 1e-3, torch.zeros(Kd, N, device=dev)
    pf = K.PackedFC(w, precise=False)
    db = torch.zeros(N, device=dev)
    for M in (32, 256):
        x, dy = torch.randn(M, Kd, device=dev), torch.randn(M, N, device=dev) * 1e-3
        rows = [("fused update", 20 * Kd * N, lambda: K.rmsprop_fc_fused(w, ms, x, dy, pf, 1e-4, db=db)),
                ("rmsprop_fc   ", 24 * Kd * N, lambda: K.rmsprop_fc(w, g, ms, pf, 1e-4)),
                ("wgrad bf16   ", 4 * Kd * N, lambda: K.fc_wgrad_bf16(x, dy, g, db))]
        if M <= 32:
            rows.append(("wgrad fp32   ", 4 * Kd * N, lambda: K.fc_wgrad(x, dy, g, db)))
        for name, nbytes, fn in rows:
            us = bench._graph_time(torch, fn, 20, warm=3)
            print("%dx%d M=%3d %s %8.1f us  %6.0f GB/s" % (Kd, N, M, name, us, nbytes / us / 1e3))
