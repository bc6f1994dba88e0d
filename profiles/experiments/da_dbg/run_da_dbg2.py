"""Diagnostic runner 2 for the distortion-aware data-gradient defect (a15): the RAW operands of the blend (8 weights + 8 x 8
source values per (pixel, 8-channel) item and round) and the source offsets the gather computed, quiet launch vs launch
beside the wide conv tile.  lib built with -DHDRSKY_DA_DEBUG2 (global-memory kernel da_conv_kernel<false,4,2,8>).
Usage: python run_da_dbg2.py <lib.so> [--runs N]"""
import argparse, importlib, os, sys
ap = argparse.ArgumentParser(); ap.add_argument("lib"); ap.add_argument("--runs", type=int, default=8)
args = ap.parse_args()
os.environ["HDRSKY_DA_REGION"] = "0"; os.environ["HDRSKY_TILE_WIDE"] = "2,4,4,2,32,1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import torch, ctypes
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"
L = importlib.import_module(PKG + "._lib"); L.LIB_PATH = os.path.abspath(args.lib)
K = importlib.import_module(PKG + ".kernels"); L.load()
dev = torch.device("cuda:0"); torch.manual_seed(0)
side = torch.cuda.Stream()
xn = torch.randn(16, 64, 256, 64, device=dev); pwn = K.PackedConv(torch.randn(4, 4, 64, 128, device=dev) * 0.03, False); bn = torch.zeros(128, device=dev)
def neighbour(n=6):
    with torch.cuda.stream(side):
        for _ in range(n): K.conv2d(xn, pwn, bn, stride=2)
B, H, W, F, C = 8, 128, 512, 32, 64
table = K.da_transpose_table(H, W, 3, 1, True, dev); dd2 = torch.randn(B, H, W, F, device=dev)
pwT = K.PackedConv(torch.randn(3, 3, C, F, device=dev) / 24, False, transpose_flip=True)
NT, IMAX, NR, WG = 256, 2, 5, 2048
osz = WG * NR * IMAX * NT * 8; rsz = WG * NR * IMAX * NT * 72
dbg = torch.zeros(osz + rsz, dtype=torch.int32, device=dev)
lib_c = ctypes.CDLL(L.LIB_PATH); lib_c.hdrsky_debug_da_set.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib_c.hdrsky_debug_da_set(dbg.data_ptr(), osz)
def run():
    dbg.zero_()
    y = K.da_conv2d_dgrad(dd2, pwT, table, 3, K.BF16); torch.cuda.synchronize()
    return y, dbg.clone()
ref, dref = run()
y2, d2 = run()
print("quiet rerun: output equal %s, dump equal %s" % (bool(torch.equal(ref, y2)), bool(torch.equal(dref, d2))), flush=True)
Oref = dref[:osz].view(WG, NR, IMAX, NT, 8); Rref = dref[osz:].view(torch.float32).view(WG, NR, IMAX, NT, 72)
dyf = dd2.view(B, H * W * F)
for r in range(args.runs):
    neighbour()
    y, d = run()
    nbad = int((y != ref).sum())
    if nbad == 0:
        print("run %d identical" % r, flush=True); continue
    O = d[:osz].view(WG, NR, IMAX, NT, 8); R = d[osz:].view(torch.float32).view(WG, NR, IMAX, NT, 72)
    do = (O != Oref).nonzero(); dr = (R.view(torch.int32) != Rref.view(torch.int32)).nonzero()
    print("run %d: %d wrong outputs; dumped (first 2048 workgroups): OFFSET words differ %d, RAW operand words differ %d" % (r, nbad, do.shape[0], dr.shape[0]), flush=True)
    if dr.shape[0] == 0: continue
    items = torch.unique(dr[:, :4], dim=0)
    print("   items (wg, t, it, tid) with differing operands: %d; its %s; lanes %s; rounds %s" % (items.shape[0], sorted(set(items[:, 2].tolist())),
          sorted(set((items[:, 3] % 64).tolist())), sorted(set(items[:, 1].tolist()))))
    comp = dr[:, 4]
    print("   components hit: weights (0-7): %d words; sources: %d words; per source k: %s" % (int((comp < 8).sum()), int((comp >= 8).sum()),
          [int((((comp - 8) // 8) == k).logical_and(comp >= 8).sum()) for k in range(8)]))
    stale = zero = 0; shown = 0
    for (wg, t, it, tid) in items[:400].tolist():
        g, b_ = Rref[wg, t, it, tid], R[wg, t, it, tid]
        og, ob = Oref[wg, t, it, tid], O[wg, t, it, tid]
        ks = sorted(set(((torch.nonzero(g.view(torch.int32) != b_.view(torch.int32)).flatten() - 8) // 8).tolist()))
        prev = Rref[wg, t - 1, it, tid] if t > 0 else None
        for k in ks:
            if k < 0: continue
            bv = b_[8 + 8 * k: 16 + 8 * k]
            if prev is not None and torch.equal(bv, prev[8 + 8 * k: 16 + 8 * k]): stale += 1
            if float(bv.abs().max()) == 0.0: zero += 1
        if shown < 6:
            shown += 1
            bsmp = wg // (H * W // 64)
            print("   item wg %d t %d it %d tid %d (lane %d): sources differing %s; offsets good %s bad %s" % (wg, t, it, tid, tid % 64, ks, og.tolist(), ob.tolist()))
            for k in ks[:3]:
                if k < 0:
                    print("      weights good %s bad %s" % (g[:8].tolist(), b_[:8].tolist())); continue
                gv, bv = g[8 + 8 * k: 16 + 8 * k], b_[8 + 8 * k: 16 + 8 * k]
                print("      k=%d good %s" % (k, ["%.4f" % v for v in gv.tolist()]))
                print("           bad  %s" % (["%.4f" % v for v in bv.tolist()]))
                # where in dy (this sample) do the bad values live?
                hit = (dyf[bsmp] == bv[0]).nonzero().flatten()
                cand = [int(h) for h in hit.tolist() if h + 8 <= dyf.shape[1] and torch.equal(dyf[bsmp, h:h + 8], bv)]
                print("           the bad 8 values are dy[sample %d] at element offsets %s (the good offset is %d)" % (bsmp, cand[:4], int(og[k])))
    print("   of the differing sources in the first 400 items: equal to the PREVIOUS round's register content %d, all-zero %d" % (stale, zero), flush=True)
    break
