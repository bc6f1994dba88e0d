"""Diagnostic runner for the distortion-aware data-gradient defect (a15): is hdrsky_da_conv2d_dgrad bit-reproducible beside a
neighbour on another stream?  Usage: python run_da_dbg.py <lib.so> [--region 0|1] [--neighbour wide|conv|gemm|stream|none]
[--dbg] [--runs N].  With --dbg (lib built with -DHDRSKY_DA_DEBUG, global-memory kernel) the per-round hashes of the A-tile
values every thread WROTE and every MFMA lane READ are compared between the quiet and the contended launches."""
import argparse, importlib, os, sys, collections
ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--region", type=int, default=0)
ap.add_argument("--neighbour", default="wide")
ap.add_argument("--dbg", action="store_true")
ap.add_argument("--runs", type=int, default=6)
ap.add_argument("--shape", default="hires")     # hires: 8x128x512 32->64 ; lowres: 32x8x32 128->128
args = ap.parse_args()
os.environ["HDRSKY_DA_REGION"] = "2" if args.region else "0"
if args.neighbour == "wide":
    os.environ["HDRSKY_TILE_WIDE"] = "2,4,4,2,32,1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import torch, ctypes
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
L = importlib.import_module(PKG + "._lib")
L.LIB_PATH = os.path.abspath(args.lib)
K = importlib.import_module(PKG + ".kernels")
lib = L.load()
dev = torch.device("cuda:0"); torch.manual_seed(0)
side = torch.cuda.Stream()
xn = torch.randn(16, 64, 256, 64, device=dev); pwn = K.PackedConv(torch.randn(4, 4, 64, 128, device=dev) * 0.03, False); bn = torch.zeros(128, device=dev)
ga = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16); gb = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
big = torch.randn(1 << 26, device=dev)
def neighbour(n=6):
    if args.neighbour == "none": return
    with torch.cuda.stream(side):
        for _ in range(n):
            if args.neighbour in ("wide", "conv"): K.conv2d(xn, pwn, bn, stride=2)
            elif args.neighbour == "gemm": torch.mm(ga, gb)
            else: big.mul_(1.0001)
if args.shape == "hires":
    B, H, W, F, C = 8, 128, 512, 32, 64
else:
    B, H, W, F, C = 32, 8, 32, 128, 128
table = K.da_transpose_table(H, W, 3, 1, True, dev); dd2 = torch.randn(B, H, W, F, device=dev)
pwT = K.PackedConv(torch.randn(3, 3, C, F, device=dev) / 24, False, transpose_flip=True)
NT, IMAX, CB = 256, 2, 2
dbg = None
if args.dbg:
    nwg = B * ((H * W + 63) // 64); nr = 5
    wsz = nwg * nr * IMAX * NT; rsz = nwg * nr * CB * 4 * NT
    dbg = torch.zeros(wsz + rsz, dtype=torch.int32, device=dev)
    lib_c = ctypes.CDLL(L.LIB_PATH)
    lib_c.hdrsky_debug_da_set.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib_c.hdrsky_debug_da_set(dbg.data_ptr(), wsz)
def run():
    if dbg is not None: dbg.zero_()
    y = K.da_conv2d_dgrad(dd2, pwT, table, 3, K.BF16); torch.cuda.synchronize()
    return y, (dbg.clone() if dbg is not None else None)
ref, dref = run()
y2, d2 = run()
print("lib %s region %d neighbour %s shape %s: quiet rerun equal: %s" % (os.path.basename(args.lib), args.region, args.neighbour, args.shape, bool(torch.equal(ref, y2))), flush=True)
nbad_runs = 0
for r in range(args.runs):
    neighbour()
    y, d = run()
    bad = (y != ref).nonzero()
    if bad.shape[0] == 0:
        print("  run %d: identical" % r, flush=True); continue
    nbad_runs += 1
    b, h, w, c = bad[:, 0], bad[:, 1], bad[:, 2], bad[:, 3]
    pit = ((h * W + w) % 64)
    key = collections.Counter(zip(b.tolist(), ((h * W + w) // 64).tolist()))
    dd = (y - ref)[b, h, w, c]
    print("  run %d: %d wrong elements in %d (sample, tile) groups; samples %s; pixel-in-tile %s; channels %d distinct; |diff| med %.3e max %.3e (ref med %.3f)" %
          (r, bad.shape[0], len(key), sorted(set(b.tolist())), sorted(set(pit.tolist())), len(set(c.tolist())), float(dd.abs().median()), float(dd.abs().max()),
           float(ref[b, h, w, c].abs().median())), flush=True)
    if d is not None:
        wsz_ = wsz
        dw = (d[:wsz_] != dref[:wsz_]).nonzero().flatten()
        dr = (d[wsz_:] != dref[wsz_:]).nonzero().flatten()
        print("     hashes: WRITTEN differ %d, READ differ %d" % (dw.numel(), dr.numel()), flush=True)
        def dec_w(i):
            tid = i % NT; i //= NT; it = i % IMAX; i //= IMAX; t = i % nr; wg = i // nr
            return dict(wg=wg, b=wg // (H * W // 64), tile=wg % (H * W // 64), t=t, it=it, tid=tid, wave=tid // 64, lane=tid % 64, m=(it * NT + tid) // 8)
        def dec_r(i):
            tid = i % NT; i //= NT; mi = i % 4; i //= 4; cb = i % CB; i //= CB; t = i % nr; wg = i // nr
            return dict(wg=wg, b=wg // (H * W // 64), tile=wg % (H * W // 64), t=t, cb=cb, mi=mi, tid=tid, wave=tid // 64, lane=tid % 64, row=mi * 16 + (tid % 16), kq=(tid % 64) // 16)
        for i in dw[:12].tolist(): print("       W", dec_w(i))
        if dw.numel():
            allw = [dec_w(i) for i in dw.tolist()[:200000]]
            print("       W summary: rounds %s its %s lanes %s waves %s rows %s" % (sorted(set(e["t"] for e in allw)), sorted(set(e["it"] for e in allw)),
                  sorted(set(e["lane"] for e in allw)), sorted(set(e["wave"] for e in allw)), sorted(set(e["m"] for e in allw))))
        for i in dr[:12].tolist(): print("       R", dec_r(i))
        if dr.numel():
            allr = [dec_r(i) for i in dr.tolist()[:200000]]
            print("       R summary: rounds %s cb %s mi %s rows %s kq %s waves %s" % (sorted(set(e["t"] for e in allr)), sorted(set(e["cb"] for e in allr)),
                  sorted(set(e["mi"] for e in allr)), sorted(set(e["row"] for e in allr)), sorted(set(e["kq"] for e in allr)), sorted(set(e["wave"] for e in allr))))
            # do the wrong outputs' (sample, tile) groups coincide with the groups whose hashes differ?
            gr = set((e["b"], e["tile"]) for e in allr); gw_ = set((e["b"], e["tile"]) for e in (allw if dw.numel() else []))
            print("       groups: output %d, READ %d (common %d), WRITTEN %d (common %d)" % (len(key), len(gr), len(gr & set(key)), len(gw_), len(gw_ & set(key))))
print("RESULT lib %s region %d neighbour %s shape %s: %d of %d contended runs differ" % (os.path.basename(args.lib), args.region, args.neighbour, args.shape, nbad_runs, args.runs), flush=True)
