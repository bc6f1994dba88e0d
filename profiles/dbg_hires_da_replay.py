import sys, os, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_hires_gpu as th
dev = torch.device("cuda:0")
da = "res,decoders"
B = 8
ldr, hdr = th._inputs(B, seed=43)
cmf, gt, cams = th._sun_inputs(B, seed=19)
d = lambda a: torch.from_numpy(a).to(dev)
args = (d(ldr), d(hdr), d(gt)); kw = dict(cmf=d(cmf), cams=[d(c) for c in cams])
def run(mode):
    tr, _ = th._hires_trainer(dev, "BF16", da)
    if mode == "eager":
        tr.step(*args, update=False, **kw)
    else:
        tr.capture(*args, **kw); tr.replay(update=False)
    torch.cuda.synchronize()
    return tr
a, b, c = run("eager"), run("eager"), run("replay")
for label, x, y in (("eager vs eager", a, b), ("eager vs replay", a, c)):
    rows = []
    for fp in ("gs", "ds"):
        X, Y = getattr(x, fp), getattr(y, fp)
        for name, (o, n, _) in X.offsets.items():
            if o + n > X.grad.numel(): continue
            e = float((X.grad[o:o+n] - Y.grad[o:o+n]).abs().max()); s = float(X.grad[o:o+n].abs().max())
            if e > 0: rows.append((e / (s + 1e-30), name, e, s))
    rows.sort(reverse=True)
    print(label, len(rows), "differing tensors"); [print("   %.3e %s (abs %.3e of %.3e)" % r) for r in rows[:6]]
    big = sorted([r for r in rows if r[3] > 1e-3], reverse=True)
    print("  tensors with |g|max > 1e-3:", len(big)); [print("   %.3e %s (abs %.3e of %.3e)" % r) for r in big[:8]]
    for nm in ("y_lin", "dyl", "dres", "dx_enc", "c3"):
        if nm in x._T and nm in y._T and torch.is_tensor(x._T[nm]):
            print("   T[%s] max abs diff %.3e of %.3e" % (nm, float((x._T[nm].float() - y._T[nm].float()).abs().max()), float(x._T[nm].float().abs().max())))
    print("  losses", x.losses.tolist(), y.losses.tolist())


def walk(o, path, out):
    if torch.is_tensor(o):
        out.append((path, o))
    elif isinstance(o, dict):
        for k, v in o.items(): walk(v, path + "." + str(k), out)
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o): walk(v, path + "[%d]" % i, out)
    elif hasattr(o, "__dict__") and not callable(o):
        for k, v in vars(o).items(): walk(v, path + "." + k, out)


print("---- tensors of the step's working set that differ between the two eager runs (insertion order)")
ta, tb = [], []
walk(a._T, "T", ta); walk(b._T, "T", tb)
for (pa, xa), (pb, xb) in zip(ta, tb):
    if pa != pb or xa.shape != xb.shape or not xa.is_floating_point(): continue
    dlt = float((xa.float() - xb.float()).abs().max())
    if dlt > 0: print("   %-50s %s diff %.3e of %.3e" % (pa, tuple(xa.shape), dlt, float(xa.float().abs().max())))
