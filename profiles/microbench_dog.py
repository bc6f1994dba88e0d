import importlib, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels")
HK = importlib.import_module(PKG + ".hooks")      # HDRSKY_* variables are read once: reload() after every change
dev = torch.device("cuda:0")
for shape in [(2, 8, 16, 3), (3, 32, 128, 3), (2, 5, 7, 3), (1, 4, 4, 1), (2, 9, 33, 2), (32, 32, 128, 3), (8, 128, 512, 3)]:
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.uniform(0, 3, shape).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(0, 3, shape).astype(np.float32)).to(dev)
    out = {}
    for fused in ("1", "0"):
        os.environ["HDRSKY_DOG_FUSED"] = fused; HK.reload()
        slot, dy = torch.zeros(1, device=dev), torch.zeros(shape, device=dev)
        K.dog_loss(a, b, 1000.0, slot, dy)
        out[fused] = (float(slot), dy)
    d = (out["1"][1] - out["0"][1]).abs()
    print(shape, "loss", out["1"][0], out["0"][0], "max|d|", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel(), "max|g|", float(out["0"][1].abs().max()))
    def tm(f):
        os.environ["HDRSKY_DOG_FUSED"] = f; HK.reload()
        slot, dy = torch.zeros(1, device=dev), torch.zeros(shape, device=dev)
        for _ in range(3): K.dog_loss(a, b, 1000.0, slot, dy)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): K.dog_loss(a, b, 1000.0, slot, dy)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 20
    print("   us per call: fused %.1f staged %.1f" % (tm("1"), tm("0")))
