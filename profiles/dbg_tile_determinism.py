"""Debug: is a conv launch on a given tile bit-reproducible run to run (outputs and statistics partials)?"""
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
torch.manual_seed(0)
# (name, B, H, W, Cin, Cout, k, stride, bf16 in/out, want_stats, affine input transform)
CASES = [("vgg2_1", 4, 64, 256, 64, 128, 3, 1, True, False, False), ("vgg2_2", 4, 64, 256, 128, 128, 3, 1, True, False, False),
         ("dis.d2 s2", 16, 64, 256, 64, 128, 4, 2, False, True, True), ("res 128", 8, 32, 128, 128, 128, 3, 1, False, True, True),
         ("d4 256->512", 16, 16, 64, 256, 512, 4, 1, False, True, True)]
for tile in ("2,4,4,2,32,1", "2,4,4,1,32,1"):
    _hook("HDRSKY_TILE_WIDE", tile)
    for name, B, H, W, Cin, Cout, k, st, b16, ws, aff in CASES:
        x = torch.randn(B, H, W, Cin, device=dev)
        if b16: x = x.to(torch.bfloat16)
        w = torch.randn(k, k, Cin, Cout, device=dev) / (k * k * Cin) ** 0.5
        pw = K.PackedConv(w, False); bias = torch.randn(Cout, device=dev)
        xf = K.InXf(mode=L.IN_AFFINE, slope=0.3, scale=torch.rand(Cin, device=dev) + 0.5, shift=torch.randn(Cin, device=dev)) if aff else None
        ref = None; bad = 0
        side = torch.cuda.Stream()
        for it in range(12):
            with torch.cuda.stream(side):                      # a neighbour on another stream
                junk = torch.randn(1 << 22, device=dev).sum()
            y, stt = K.conv2d(x, pw, bias, stride=st, xf=xf, want_stats=ws, out_bf16=b16, out_slope=0.0 if b16 else 1.0)
            torch.cuda.synchronize()
            snap = (y.clone(), stt.part.clone() if stt is not None else None)
            if ref is None: ref = snap; continue
            if not torch.equal(ref[0], snap[0]) or (ws and not torch.equal(ref[1], snap[1])): bad += 1
        print("%-14s tile %-14s %s   %s" % (name, tile, K.conv_kernel_name(K.conv_desc(B, H, W, Cin, Cout, k, k, st, True, 1)).replace("conv_igemm_kernel", ""), "REPRODUCIBLE" if bad == 0 else "%d of 11 runs DIFFER" % bad), flush=True)
