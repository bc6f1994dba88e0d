#!/usr/bin/env python3
"""Tile table of conv_igemm_kernel re-measured in the regime the training step runs in: every conv / data-gradient launch of one
eager bench-mode step is recorded (kernels.conv2d wrapped), identical launches are grouped, and each group is timed under every
instantiated tile shape (HDRSKY_TILE hook) alone AND saturated - three HIP streams each replaying a hipGraph of N copies of the
launch, i.e. the chip full of waves of its own kind.  The table of rounds 1-3 was tuned on the first number; the three-stream step
pays the second (profiles/r05_conv_throughput.txt).  Prints per group: launches per step, the table's tile, the best saturated tile.
usage: python profiles/tile_sweep.py [--n 30]"""
import argparse, importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HDRSKY_EXPERIMENTS"] = "1"
import bench
P, synth, trainer, K, HK, L = (importlib.import_module(bench.PKG + "." + m) for m in ("params", "synth", "trainer", "kernels", "hooks", "_lib"))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=30)
ap.add_argument("--hires", action="store_true", help="the 128x512 training step at batch 8 (sun-pose net external: configs[4] on one GPU)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
if args.hires:
    H, W, B = 128, 512, 8
    nets = [P.init_params(P.generator_spec(H, W), 0), None, P.init_params(P.discriminator_spec(), 2), P.init_params(P.vgg_spec(), 3)]
    g = torch.Generator(device=dev); g.manual_seed(1234)
    ldr = torch.round(torch.rand(B, H, W, 3, device=dev, generator=g) * 255.0) / 255.0
    hdr = ldr ** 2.2 * (1.0 + 3.0 * torch.rand(B, H, W, 3, device=dev, generator=g))
    cmf = torch.softmax(3.0 * torch.randn(B, H * W, device=dev, generator=g), dim=1).contiguous()
    gt = torch.softmax(3.0 * torch.randn(B, H * W, device=dev, generator=g), dim=1).contiguous()
    cams = [torch.relu(torch.randn(B, H >> i, W >> i, 1, device=dev, generator=g)).contiguous() for i in range(3)]
    tr = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16, im_height=H, im_width=W, sunpose="external")
    step = lambda: tr.step(ldr, hdr, gt, update=False, cmf=cmf, cams=cams)
else:
    nets = [P.init_params(P.generator_spec(), 0), P.init_params(P.sunpose_spec(), 1), P.init_params(P.discriminator_spec(), 2), P.init_params(P.vgg_spec(), 3)]
    b = synth.make_batch(32, seed=1234)
    ldr, hdr, gt = (torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt"))
    tr = trainer.Trainer(*nets, device=dev, precise=False, compute=K.BF16)
    step = lambda: tr.step(ldr, hdr, gt, update=False)
step()
torch.cuda.synchronize()
calls = []
orig = K.conv2d


def rec(x, pw, bias=None, **kw):
    if kw.get("pair") is None:      # (the paired launches take the tile of one of their layers: the unpaired rows stand for them)
        calls.append((x, pw, bias, dict(kw), K._LABEL[0]))
    return orig(x, pw, bias, **kw)


K.conv2d = rec
trainer.K.conv2d = rec
step()
torch.cuda.synchronize()
K.conv2d = orig
trainer.K.conv2d = orig


def sig(c):
    x, pw, bias, kw, _ = c
    d = kw.get("desc")
    xf = kw.get("xf")
    return (tuple(x.shape), str(x.dtype), pw.KH, pw.KW, pw.Cout, pw.flip, kw.get("stride", 1), kw.get("upsample", 1), xf.mode if xf else 0,
            bool(kw.get("want_stats")), bool(kw.get("out_bf16")), kw.get("residual") is not None, kw.get("mask_bf16") is not None,
            (d.dilate, d.Ho, d.Wo) if d is not None else None, bool(kw.get("emit_xb")))


groups = {}
for c in calls:
    g = groups.setdefault(sig(c), {"call": c, "n": 0, "labels": []})
    g["n"] += 1
    if c[4] and c[4] not in g["labels"]:
        g["labels"].append(c[4])
TILES = [None, "2,4,4,1,32,1", "2,2,4,2,32,1", "2,4,4,2,32,1", "1,4,4,1,32,1", "2,4,2,1,32,1", "4,2,4,1,32,1", "1,8,4,1,32,1", "8,1,4,1,32,1", "4,1,4,1,32,1", "2,2,4,1,32,1",
         "2,2,4,2,32,0", "4,1,4,2,32,0", "8,1,4,2,32,0", "4,2,2,2,32,0", "1,8,2,1,16,1", "1,4,4,1,16,1", "2,4,2,1,16,0",
         "4,1,4,4,32,1", "2,2,4,4,32,1", "4,1,2,4,32,1", "1,4,4,4,32,1"]
streams = [torch.cuda.Stream() for _ in range(3)]


def run(graphs, reps=3):
    torch.cuda.synchronize()
    for g, s in graphs:
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, s in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


total_table = total_best = 0.0
for key, g in sorted(groups.items(), key=lambda kv: -kv[1]["n"]):
    x, pw, bias, kw, _ = g["call"]
    kw = {k: v for k, v in kw.items() if k not in ("out", "emit_xb")}      # fresh outputs; the emit form has its own instantiations (EMIT)
    d = kw.get("desc")
    Ho, Wo = (d.Ho, d.Wo) if d is not None else (None, None)
    res = {}
    for tile in TILES:
        if tile is None:
            os.environ.pop("HDRSKY_TILE", None)
        else:
            if int(tile.split(",")[4]) == 16 and (x.shape[2] >= 32):
                continue
            os.environ["HDRSKY_TILE"] = tile
        HK.reload()
        fn = lambda: orig(x, pw, bias, **kw)
        try:
            fn(); torch.cuda.synchronize()
        except Exception:
            continue
        gs = []
        for s in streams:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(args.n):
                    fn()
            gs.append((gr, s))
        res[tile or "table"] = (run(gs[:1]) / args.n, run(gs) / args.n)
        del gs
    os.environ.pop("HDRSKY_TILE", None); HK.reload()
    if "table" not in res:
        continue
    best = min(res, key=lambda t: res[t][1])
    total_table += g["n"] * res["table"][1]; total_best += g["n"] * res[best][1]
    dd = d if d is not None else K.conv_desc(x.shape[0], x.shape[1], x.shape[2], x.shape[3], pw.Cout, pw.KH, pw.KW, kw.get("stride", 1), kw.get("same", True), kw.get("upsample", 1))
    dd.compute = K.BF16
    name = K.conv_kernel_name(dd)
    print("x%d %-26s %s %dx%d ->%d %s s%d up%d xf%d st%d ob%d | table %-22s alone %6.2f sat %6.2f | best sat %-14s alone %6.2f sat %6.2f (%+.0f%%) | %s" % (
        g["n"], str(tuple(x.shape)), "bf16" if x.dtype == torch.bfloat16 else "f32 ", pw.KH, pw.KW, pw.Cout, "dgrad" if pw.flip else "fwd  ", kw.get("stride", 1),
        kw.get("upsample", 1), key[8], key[9], key[10], name[18:41], res["table"][0], res["table"][1], best, res[best][0], res[best][1],
        100.0 * (res[best][1] / res["table"][1] - 1.0), "; ".join(g["labels"][:2])), flush=True)
    top = sorted(res.items(), key=lambda kv: kv[1][1])[:4]
    print("      " + "   ".join("%s: %.2f/%.2f" % (t, a, s_) for t, (a, s_) in top), flush=True)
print("sum over the step's conv launches, saturated: table %.0f us, best per group %.0f us" % (total_table, total_best))
