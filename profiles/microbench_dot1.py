import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
os.environ["HDRSKY_EXPERIMENTS"] = "1"      # HDRSKY_NO_DOT1 is a tuning hook of the C library
K = importlib.import_module(PKG + ".kernels"); L = importlib.import_module(PKG + "._lib"); HK = importlib.import_module(PKG + ".hooks")
dev = torch.device("cuda:0")
for B in (32, 64):
    x = torch.randn(B, 4, 16, 512, device=dev)
    w = torch.randn(4, 4, 512, 1, device=dev) * 0.02
    pw = K.PackedConv(w, False); bias = torch.zeros(1, device=dev)
    xf = K.InXf(mode=L.IN_AFFINE, slope=0.3, scale=torch.rand(512, device=dev) + 0.5, shift=torch.randn(512, device=dev))
    for env in ("", "1"):
        if env: os.environ["HDRSKY_NO_DOT1"] = "1"
        else: os.environ.pop("HDRSKY_NO_DOT1", None)
        HK.reload()      # the variables are read once (csrc/hooks.h): re-read after every change
        f = lambda: K.conv2d(x, pw, bias, same=False, xf=xf)
        for _ in range(3): f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print("B=%d %s: %.2f us" % (B, "igemm" if env else "dot1 ", e0.elapsed_time(e1) * 1e3 / 20))
