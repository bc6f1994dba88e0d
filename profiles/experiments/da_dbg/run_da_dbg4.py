"""Diagnostic runner 4 for the distortion-aware data-gradient defect (a15): numerical forensics.  The lib (built with
-DHDRSKY_DA_DEBUG -DHDRSKY_DA_DEBUG_SEL=4) dumps the 8 blended bf16 values every thread stores into the A tile.  The blend
is re-computed on the host from the tables and dy (float32, the kernel's order of operations) and the WRONG values of a
contended launch are matched against hypotheses: a source dropped, a weight of another source / round, stale operands."""
import argparse, importlib, os, sys, itertools
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
wsz = nwg * NR * IMAX * NT * 4; rsz = nwg * NR * CB * 4 * NT
dbg = torch.zeros(wsz + rsz, dtype=torch.int32, device=dev)
lib_c = ctypes.CDLL(L.LIB_PATH); lib_c.hdrsky_debug_da_set.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib_c.hdrsky_debug_da_set(dbg.data_ptr(), wsz)
def run():
    dbg.zero_()
    y = K.da_conv2d_dgrad(dd2, pwT, table, 3, K.BF16); torch.cuda.synchronize()
    return y, dbg[:wsz].clone().view(nwg, NR, IMAX, NT, 4)
ref, href = run()
gidx = table[0].cpu().numpy(); gw = table[1].cpu().numpy(); dy = dd2.cpu().numpy().reshape(B, H * W, F)
f32, f64 = np.float32, np.float64
def fma(a, b, c): return f32(f64(a) * f64(b) + f64(c))
def bf16_bits(v):     # round to nearest even, as v_cvt_pk_bf16_f32
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint32)
def operands(wg, t, it, tid):
    b, tile = divmod(wg, H * W // 64)
    i = it * NT + tid; m, qr = divmod(i, 8); tsub, q = divmod(qr, 4)
    tn = t * 2 + tsub; ok = tn < 9; tn = min(tn, 8)
    pix = tile * 64 + m
    gi, w = gidx[pix, tn], gw[pix, tn].copy()
    w[gi < 0] = 0.0
    if not ok: w[:] = 0.0
    src = np.stack([dy[b, max(int(g), 0), q * 8:q * 8 + 8] if g >= 0 else dy[b, 0, q * 8:q * 8 + 8] for g in gi])   # [8 sources][8 channels]
    return w.astype(np.float32), src.astype(np.float32), dict(b=b, tile=tile, m=m, q=q, tn=tn, ok=ok, gi=gi.tolist())
def blend(w, src):
    # the compiled order (ISA of da_conv_kernel<false,4,2,8>): A = w1*c1; A = fma(c0,w0,A); A = fma(c2,w2,A); A = fma(c3,w3,A);
    # B = w5*c5; B = fma(c4,w4,B); B = fma(c6,w6,B); B = fma(c7,w7,B); v = A + B
    A = f32(w[1] * src[1]); A = fma(src[0], w[0], A); A = fma(src[2], w[2], A); A = fma(src[3], w[3], A)
    Bv = f32(w[5] * src[5]); Bv = fma(src[4], w[4], Bv); Bv = fma(src[6], w[6], Bv); Bv = fma(src[7], w[7], Bv)
    return f32(A + Bv)
def unpack(words):    # 4 dwords -> 8 bf16 bit patterns
    w = np.asarray(words, dtype=np.int64) & 0xFFFFFFFF
    return np.stack([w & 0xFFFF, w >> 16], 1).reshape(-1).astype(np.uint32)
def as_f(bits): return (bits.astype(np.uint32) << 16).view(np.float32)
# sanity: the emulation reproduces the quiet launch
chk = 0; tot = 0
rng = np.random.default_rng(0)
for _ in range(300):
    wg, t, it, tid = int(rng.integers(nwg)), int(rng.integers(NR)), int(rng.integers(IMAX)), int(rng.integers(NT))
    w, src, _info = operands(wg, t, it, tid)
    tot += 1; chk += int(np.array_equal(bf16_bits(blend(w, src)), unpack(href[wg, t, it, tid].tolist())))
print("host emulation reproduces the quiet launch's A values on %d of %d random items" % (chk, tot), flush=True)
for r in range(args.runs):
    neighbour()
    y, h = run()
    if torch.equal(y, ref):
        print("run %d identical" % r, flush=True); continue
    items = (h != href).any(dim=-1).nonzero()
    print("run %d: %d items differ; lanes %s its %s" % (r, items.shape[0], sorted(set((items[:, 3] % 64).tolist())), sorted(set(items[:, 2].tolist()))), flush=True)
    stats = {}
    shown = 0
    for (wg, t, it, tid) in items[:600].tolist():
        w, src, info = operands(wg, t, it, tid)
        good, bad = unpack(href[wg, t, it, tid].tolist()), unpack(h[wg, t, it, tid].tolist())
        emu_ok = np.array_equal(bf16_bits(blend(w, src)), good)
        found = None
        # hypotheses
        for k in range(8):                                   # one source dropped
            w2 = w.copy(); w2[k] = 0
            if np.array_equal(bf16_bits(blend(w2, src)), bad): found = "source %d dropped (weight %.4f, gi %d)" % (k, w[k], info["gi"][k]); break
        if found is None:
            for k, k2 in itertools.permutations(range(8), 2):   # source k multiplied by the weight of source k2
                w2 = w.copy(); w2[k] = w[k2]
                if np.array_equal(bf16_bits(blend(w2, src)), bad): found = "source %d with the weight of source %d" % (k, k2); break
        if found is None:
            for k, k2 in itertools.permutations(range(8), 2):   # source k replaced by the data of source k2
                s2 = src.copy(); s2[k] = src[k2]
                if np.array_equal(bf16_bits(blend(w, s2)), bad): found = "source %d data replaced by source %d's" % (k, k2); break
        if found is None and t > 0:
            wp, sp, _ = operands(wg, t - 1, it, tid)
            for k in range(8):
                s2 = src.copy(); s2[k] = sp[k]
                if np.array_equal(bf16_bits(blend(w, s2)), bad): found = "source %d data stale (previous round)" % k; break
                w2 = w.copy(); w2[k] = wp[k]
                if np.array_equal(bf16_bits(blend(w2, src)), bad): found = "weight %d stale (previous round)" % k; break
            if found is None and np.array_equal(bf16_bits(blend(wp, sp)), bad): found = "whole item stale (previous round)"
        if found is None:
            # which channels differ, and by how much relative to single contributions?
            d = as_f(bad) - as_f(good)
            contrib = w[:, None] * src
            ratios = []
            for k in range(8):
                if abs(w[k]) > 0:
                    rr = d / np.where(np.abs(contrib[k]) > 1e-12, contrib[k], np.nan)
                    ratios.append((k, float(np.nanmedian(rr)), float(np.nanstd(rr))))
            found = "unexplained; diff/contribution per source (median, std): %s" % [(k, round(a, 3), round(s, 3)) for k, a, s in ratios]
        key = found.split(" (")[0] if found.startswith("source") else found.split(";")[0]
        stats[key] = stats.get(key, 0) + 1
        if shown < 8:
            shown += 1
            print("   item wg %d t %d it %d tid %d lane %d m %d q %d tn %d: emulation==good %s | %s" % (wg, t, it, tid, tid % 64, info["m"], info["q"], info["tn"], emu_ok, found))
            print("      w   ", ["%.4f" % v for v in w]); print("      good", ["%.4f" % v for v in as_f(good)]); print("      bad ", ["%.4f" % v for v in as_f(bad)])
    print("   hypothesis tally over %d items: %s" % (min(600, items.shape[0]), stats), flush=True)
    break
