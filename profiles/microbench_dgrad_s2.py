"""Data gradient of the stride-2 layers: output phases on the un-stuffed gradient (conv_igemm_kernel<..., PH = true>) against
the zero-stuffed operand (HDRSKY_NO_PHASE=1), the step's shapes, 20 launches per hipGraph replay.
[--tiles "wm,wn,mi,ni,tw,db;..."]: additionally time the phase form on these tiles (HDRSKY_TILE)."""
import importlib, os, sys, torch

def _hook(name, value):
    """Set / clear a HDRSKY_* variable and make the package + library read it (they read the environment once:
    hooks.py, csrc/hooks.h; tuning hooks need the HDRSKY_EXPERIMENTS=1 gate)."""
    import importlib, os, sys
    os.environ["HDRSKY_EXPERIMENTS"] = "1"
    if value is None: os.environ.pop(name, None)
    else: os.environ[name] = str(value)
    mods = [m for n, m in sys.modules.items() if n.endswith("_amd.hooks")]
    if mods: mods[0].reload()


sys.path.insert(0, os.getcwd())
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
K = importlib.import_module(PKG + ".kernels"); L = importlib.import_module(PKG + "._lib")
dev = torch.device("cuda:0")
tiles = sys.argv[sys.argv.index("--tiles") + 1].split(";") if "--tiles" in sys.argv else []
# (name, B, H, W, Cin, Cout, k, dy as bf16 + bf16 output) of the forward conv
CASES = [("dis.d3 4x4 128->256", 32, 8, 32, 128, 256, 4, True), ("dis.d3 4x4 128->256", 64, 8, 32, 128, 256, 4, True),
         ("dis.d2 4x4 64->128", 32, 16, 64, 64, 128, 4, True), ("dis.d2 4x4 64->128", 64, 16, 64, 64, 128, 4, True),
         ("dis.d1 4x4 6->64", 32, 32, 128, 6, 64, 4, False),
         ("gen.conv2_d 3x3 32->64", 32, 32, 128, 32, 64, 3, False), ("gen.conv3_d 3x3 64->128", 32, 16, 64, 64, 128, 3, False)]


def timed(f):
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    return best


for name, B, H, W, Cin, Cout, k, b16 in CASES:
    fd = K.conv_desc(B, H, W, Cin, Cout, k, k, 2, True, 1)
    w = torch.randn(k, k, Cin, Cout, device=dev) * 0.05
    pT = K.PackedConv(w, False, transpose_flip=True)
    dy = torch.randn(B, fd.Ho, fd.Wo, Cout, device=dev)
    if b16: dy = dy.to(torch.bfloat16)
    out = torch.empty(B, H, W, Cin, device=dev, dtype=torch.bfloat16 if b16 else torch.float32)
    f = lambda: K.conv2d_dgrad(dy, pT, fd, out=out, out_bf16=b16)
    flop = 2.0 * B * fd.Ho * fd.Wo * k * k * Cin * Cout
    row = []
    for env in ("1", ""):
        if env: _hook("HDRSKY_NO_PHASE", "1")
        else: _hook("HDRSKY_NO_PHASE", None)
        us = timed(f)
        row.append("%s %6.2f us %6.1f TFLOP/s %s" % ("stuffed" if env else "phases ", us, flop / us * 1e-6,
                                                     K.conv_kernel_name(K.conv_dgrad_desc(fd)).replace("conv_igemm_kernel", "")))
    for t in tiles:
        _hook("HDRSKY_TILE", t)
        try:
            row.append("phases on %s: %6.2f us" % (t, timed(f)))
        except Exception as e:
            row.append("phases on %s: %s" % (t, type(e).__name__))
        _hook("HDRSKY_TILE", None)
    print("%-26s B=%2d | %s" % (name, B, " | ".join(row)), flush=True)
