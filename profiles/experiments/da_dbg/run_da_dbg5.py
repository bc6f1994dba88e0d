"""Diagnostic runner 5 for the distortion-aware data-gradient defect (a15): the blend's fp32 results (lib built with
-DHDRSKY_DA_DEBUG -DHDRSKY_DA_DEBUG_SEL=5), quiet launch vs launch beside the wide conv tile.  With the exact fp32 values the
altered term can be solved for: for a failing (item, channel j), delta = bad - good; for every source k with a non-zero
weight, the operand value X_k the blend must have read instead of c_k[j] is c_k[j] + delta / w_k - matched against zero,
the other channels / sources of the item, the previous round's registers and the gather's integer temporaries."""
import argparse, importlib, os, sys, collections
ap = argparse.ArgumentParser(); ap.add_argument("lib"); ap.add_argument("--runs", type=int, default=8)
args = ap.parse_args()
os.environ["HDRSKY_DA_REGION"] = "0"; os.environ["HDRSKY_TILE_WIDE"] = "2,4,4,2,32,1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes
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
NT, IMAX, NR, CB = 256, 2, 5, 2
nwg = B * (H * W // 64)
wsz = nwg * NR * IMAX * NT * 8; rsz = nwg * NR * CB * 4 * NT
dbg = torch.zeros(wsz + rsz, dtype=torch.int32, device=dev)
lib_c = ctypes.CDLL(L.LIB_PATH); lib_c.hdrsky_debug_da_set.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib_c.hdrsky_debug_da_set(dbg.data_ptr(), wsz)
def run():
    dbg.zero_()
    y = K.da_conv2d_dgrad(dd2, pwT, table, 3, K.BF16); torch.cuda.synchronize()
    return y, dbg[:wsz].clone().view(torch.float32).view(nwg, NR, IMAX, NT, 8)
ref, href = run()
gidx = table[0].cpu().numpy(); gw = table[1].cpu().numpy(); dy = dd2.cpu().numpy().reshape(B, H * W, F)
def operands(wg, t, it, tid):
    b, tile = divmod(wg, H * W // 64)
    i = it * NT + tid; m, qr = divmod(i, 8); tsub, q = divmod(qr, 4)
    tn = t * 2 + tsub; ok = tn < 9; tn = min(tn, 8)
    pix = tile * 64 + m
    gi, w = gidx[pix, tn], gw[pix, tn].copy()
    w[gi < 0] = 0.0
    if not ok: w[:] = 0.0
    src = np.stack([dy[b, max(int(g), 0), q * 8:q * 8 + 8] if g >= 0 else dy[b, 0, q * 8:q * 8 + 8] for g in gi])
    return w.astype(np.float32), src.astype(np.float32), dict(b=b, tile=tile, m=m, q=q, tn=tn, ok=ok, gi=gi.tolist())
for r in range(args.runs):
    neighbour()
    y, h = run()
    if torch.equal(y, ref):
        print("run %d identical" % r, flush=True); continue
    diff = (h.view(torch.int32) != href.view(torch.int32))
    items = diff.any(dim=-1).nonzero()
    chan = collections.Counter(tuple(diff[tuple(i)].nonzero().flatten().tolist()) for i in items[:3000].tolist())
    print("run %d: %d items differ; lanes %s its %s rounds %s; channels that differ: %s" % (r, items.shape[0], sorted(set((items[:, 3] % 64).tolist())),
          sorted(set(items[:, 2].tolist())), sorted(set(items[:, 1].tolist())), dict(chan)), flush=True)
    tally = collections.Counter(); shown = 0
    for (wg, t, it, tid) in items[:1500].tolist():
        w, src, info = operands(wg, t, it, tid)
        good, bad = href[wg, t, it, tid].cpu().numpy(), h[wg, t, it, tid].cpu().numpy()
        wp, sp, _ = operands(wg, t - 1, it, tid) if t > 0 else (None, None, None)
        for j in np.nonzero(good != bad)[0].tolist():
            delta = np.float64(bad[j]) - np.float64(good[j])
            expl = []
            for k in range(8):
                if w[k] == 0: continue
                X = np.float64(src[k, j]) + delta / np.float64(w[k])          # what the blend must have read as c_k[j]
                tol = 4e-6 * (abs(good[j]) + abs(w[k] * src[k, j])) / abs(w[k]) + 1e-6
                if abs(X) <= tol: expl.append("c%d[%d] read as 0" % (k, j))
                for k2 in range(8):
                    for j2 in range(8):
                        if (k2, j2) != (k, j) and abs(X - src[k2, j2]) <= tol: expl.append("c%d[%d] read as c%d[%d]" % (k, j, k2, j2))
                if sp is not None:
                    for k2 in range(8):
                        for j2 in range(8):
                            if abs(X - sp[k2, j2]) <= tol and abs(sp[k2, j2] - src[k2, j2]) > tol: expl.append("c%d[%d] read as the PREVIOUS round's c%d[%d]" % (k, j, k2, j2))
                # a weight altered instead: w_k read as Y
                if src[k, j] != 0:
                    Y = np.float64(w[k]) + delta / np.float64(src[k, j])
                    tolw = 4e-6 * (abs(good[j]) + abs(w[k] * src[k, j])) / abs(src[k, j]) + 1e-7
                    if abs(Y) <= tolw: expl.append("w%d read as 0 (channel %d)" % (k, j))
                    for k2 in range(8):
                        if k2 != k and w[k2] != 0 and abs(Y - w[k2]) <= tolw: expl.append("w%d read as w%d (channel %d)" % (k, k2, j))
            key = "; ".join(sorted(set(e.split(" (")[0] for e in expl))) if expl else "unexplained"
            # normalise the channel out of the key for the tally
            tally[key] += 1
            if shown < 10:
                shown += 1
                print("   wg %d t %d tid %d lane %d ch %d: good %.7f bad %.7f delta %.3e | w %s | c[:,j] %s | %s" % (wg, t, tid, tid % 64, j, good[j], bad[j], delta,
                      ["%.4f" % v for v in w[:4]], ["%.4f" % v for v in src[:4, j]], key))
    print("   tally over the first 1500 items:", dict(tally.most_common(12)), flush=True)
    break
