#!/usr/bin/env python3
"""Throughput bench of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1: in this process; N > 1: this process starts the N ranks
                                                            as a child torchrun job and relays rank 0's line)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W      (one rank per GPU, RCCL)
A rank count that differs from --gpus (WORLD_SIZE, visible devices) is fatal: the line never reports an n_gpus it did not run on.

One "step" = one pass of the workload over one batch of synthetic sky panoramas per GPU (inputs resident in HBM before the
timed region; the pass is replayed as hipGraphs).  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  all (default)   = the whole BASELINE metric in one line: `train` timed first (metric / value / ms_per_step), then `fwd`
                    (object "fwd": ms_per_step, ms_per_img, images_per_s, algorithmic_tflops), plus "roofline" (the
                    dominant MFMA kernel timed live), "roofline_hbm" (the HBM-bound kernels timed live against 8 TB/s),
                    "parity" (bf16 bench mode vs the fp32-class BF16X3 mode on the bench batch) and "cpu_baseline".
  train           = configs[2]/[3]: the full train.py step - generator + sun-pose + Grad-CAM + sun-radiance forward,
                    discriminator x3, VGG16 perceptual, DoG/L1/KL/LSGAN losses, both backward passes, RMSprop x2,
                    weight re-packing - batch 32 per GPU; N > 1: data parallel, gradients all-reduced (RCCL).
  fwd             = configs[1]: generator + sun-pose net (+ Grad-CAM sweep) forward only, batch 32 (replicas).
  hires           = configs[4] on one GPU: 128x512 panoramas, 8 per GPU - generator encoder with plain AND with
                    distortion-aware res blocks, both decoders, blending, discriminator, VGG16 x2 and the L1 / DoG /
                    perceptual / LSGAN loss values (forward + losses).  The faithful 128x512 sun-pose net has 12.9 G
                    parameters: it is not part of the step; its first Dense layer is timed
                    separately as an HBM-bound weight-streaming GEMM slice ("sunpose_fc").
  hires-train     = configs[4] as the metric defines it ("training images/s") on one GPU: the full train.py step at
                    128x512, 8 per GPU - generator (+ sun-radiance head) forward, discriminator x3, VGG16 perceptual,
                    DoG / L1 / LSGAN losses, both backward passes, RMSprop x2, re-packing - with the same SUBSTITUTION:
                    the sun-pose net's outputs (cmf + three Grad-CAM maps) are inputs of the step
                    (Trainer(sunpose="external")).  --da PARTS selects distortion-aware res blocks / decoders.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd"

MFMA_PEAK_TFLOPS = 2500.0    # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0        # HBM3E spec, same table (6.3 TB/s is what a float4 copy reaches)
FWD_MFLOP_PER_IMG = 3220.3   # G + S + C (algorithmic 2*MAC of conv/dense contractions)
TRAIN_MFLOP_PER_IMG = 16900.0  # 3(G+S) + C + 8D + 3V
# SURVEY.md section 8d at 128x512: generator incl. sun-radiance head G = 33 862.7, discriminator D = 6 656.8 per call,
# VGG16 V = 24 385.7 per call; train step without the sun-pose net = 3 G + 8 D + 3 V
HIRES_TRAIN_MFLOP_PER_IMG = 3 * 33862.7 + 8 * 6656.8 + 3 * 24385.7
HIRES_MFLOP_PER_IMG = 2 * 16320.0 + 10900.0 + 6657.0 + 2 * 24390.0   # two encoders + decoders + discriminator + VGG x2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (configs[1..3]: 32, configs[4]: 8)")
    ap.add_argument("--workload", default="all", choices=["all", "train", "fwd", "hires", "hires-train"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity object (its 1 200 training steps would dominate a profile)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-roofline-top", action="store_true", help="skip the per-layer roofline table of the training step")
    ap.add_argument("--roofline-rows", type=int, default=5, help="rows of roofline_top (0 = every traced group)")
    ap.add_argument("--da", nargs="?", const="res", default="", metavar="PARTS",
                    help="train / fwd workload with distortion-aware layers, forward and backward: comma list of res "
                         "(distortion_aware_ops.conv2d in the res blocks, generator.py:14,18's commented-out variant; the "
                         "default of a bare --da), sunpose (sunpose_net.py:11,16), decoders (distortion_aware_ops.deconv2d in "
                         "both decoders), or all")
    ap.add_argument("--dp-mode", default=None, help="gradient exchange of the N > 1 training step (parallel.py: MODES)")
    ap.add_argument("--defer-dense", action="store_true",
                    help="train workload: the Dense kernels' update opens the NEXT replay (Trainer(defer_dense=True), flushed inside the timed "
                         "region) instead of closing its own step - measured level with the default (profiles/r05_defer_dense_ab.txt)")
    ap.add_argument("--steps-only", action="store_true",
                    help="only the timed loop of the chosen workload: no roofline / roofline_top / roofline_hbm / fp32_class / parity / "
                         "cpu_baseline legs (the command behind profiles/r05_train_b32_kernel_stats.csv: a profile of it holds "
                         "bench-mode steps and nothing else)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the timed loop of the roofline kernel and print its object (the command behind "
                         "profiles/r02_roofline_kernel_stats.csv)")
    args = ap.parse_args()
    if args.steps_only:
        args.no_cpu_baseline = args.no_parity = args.no_roofline_top = True
    return args


def _graph_time(torch, launch, iters, warm=10):
    """Average launch-to-launch time (us) of `launch` replayed `iters` times from one hipGraph, HIP events on the
    launch stream."""
    for _ in range(warm):
        launch()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with _no_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(iters):
            launch()
    g.replay()
    torch.cuda.synchronize()
    e0.record(stream)
    g.replay()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def dominant_kernel_roofline(torch, K, pw, batch, h, w, iters=200):
    """The res-block convolution (3x3, 128->128 on [B,8,32,128]: 12 forward launches + their data-gradient twins per
    step) timed live with HIP events on the launch stream: the sample-resident launch of csrc/res_conv.hip in its
    training-forward form (conv + InstanceNorm + leaky, bf16 activation out, xhat / rstd saved for the backward pass).
    Algorithmic FLOPs per launch = 2*(B*8*32)*(3*3*128)*128; algorithmic bytes = x + y + xhat (bf16) + the filter."""
    dev = pw.hi.device
    x = torch.randn(batch, h // 4, w // 4, 128, device=dev).to(torch.bfloat16)
    gamma, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    us = _graph_time(torch, lambda: K.resconv_fwd(x, pw, None, gamma, beta, 0.1, save=True), iters)
    flop = 2.0 * (batch * (h // 4) * (w // 4)) * (9 * 128) * 128
    achieved = flop / (us * 1e-6) / 1e12
    # HBM-side bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE): a RECORDED value
    # (traffic_source names the file and the commit it was measured at), null when no record exists for this batch
    traffic, source = None, None
    for name in ("r04_pmc_resconv.json", "r02_pmc_resconv.json"):      # the newest record of this (unchanged since round 2) kernel
        pmc = os.path.join(ROOT, "profiles", name)
        if batch == 32 and os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f)
            traffic, source = rec.get("hbm_bytes_per_launch"), "profiles/%s @ %s" % (name, rec.get("commit", "?"))
            break
    return {"bound": "mfma", "kernel": "resconv_kernel<4> (res-block 3x3 128->128 + InstanceNorm + leaky, B=%d)" % batch,
            "role": "the res-block launch: 36 of the step's launches, ~6 % of its kernel time - see roofline_top for the "
                    "launches that cost the most",
            "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 5), "traffic": traffic, "traffic_source": source,
            "avg_launch_us": round(us, 3), "flop_per_launch": flop,
            "algorithmic_bytes": batch * 256 * 128 * 2 * 3 + 9 * 128 * 128 * 2}


def gemm1x1_roofline(torch, K, batch=8, h=32, w=128, C=128, F=128, iters=100):
    """The matmul of a distortion-aware 3x3 layer on its written gathered operand (hdrsky_gemm1x1_bf16: G [B,h,w,9C] bf16 x the
    layer's packed filter) on the res-block shape of the 128x512 network, timed live like the roofline object of the driver
    line.  Algorithmic FLOPs = 2 * B*h*w * 9C * F; algorithmic bytes = G once (bf16) + the filter + the fp32 output."""
    dev = torch.device("cuda", torch.cuda.current_device())
    G = torch.randn(batch, h, w, 9 * C, device=dev).to(torch.bfloat16)
    pw = K.PackedConv(torch.randn(3, 3, C, F, device=dev) / (9 * C) ** 0.5, False).as_1x1()
    bias = torch.zeros(F, device=dev)
    us = _graph_time(torch, lambda: K.gemm1x1(G, pw, bias, want_stats=True), iters)
    flop = 2.0 * batch * h * w * 9 * C * F
    achieved = flop / (us * 1e-6) / 1e12
    traffic, source = None, None          # HBM-side bytes per launch: a RECORDED PMC measurement of this shape (profiles/pmc_gemm1x1.py)
    pmc = os.path.join(ROOT, "profiles", "r04_pmc_gemm1x1.json")
    if (batch, h, w, C, F) == (8, 32, 128, 128, 128) and os.path.exists(pmc):
        with open(pmc) as f:
            rec = json.load(f)
        traffic, source = rec.get("hbm_bytes_per_launch"), "profiles/r04_pmc_gemm1x1.json @ %s" % rec.get("commit", "?")
    return {"bound": "mfma", "kernel": "gemm1x1_kernel<4> (distortion-aware 3x3 %d->%d layer on its written gathered operand, %dx%d maps, B=%d)" % (C, F, h, w, batch),
            "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 5),
            "traffic": traffic, "traffic_source": source, "avg_launch_us": round(us, 3), "flop_per_launch": flop,
            # the operand G is read once from HBM: against that stream the launch is much closer to its roof than against the matrix cores
            "hbm_achieved_gbs": round((batch * h * w * (9 * C * 2 + F * 4) + 9 * C * F * 2) / (us * 1e-6) / 1e9, 1), "hbm_peak_gbs": HBM_PEAK_GBS,
            "algorithmic_bytes": batch * h * w * (9 * C * 2 + F * 4) + 9 * C * F * 2}


def roofline_top(torch, K, tr, ldr, hdr, gt, top=5, iters=30):
    """The matrix-core launches of ONE training step ranked by the kernel time they cost: an eager step is traced
    (kernels.TRACE: every conv / data-gradient / weight-gradient / sample-resident launch with its layer label, kernel
    instantiation and ALGORITHMIC flop count), identical launches are grouped, each group's launch is re-issued `iters` times
    from one hipGraph and timed with HIP events on the launch stream (alone on the chip, back to back), and the `top` groups
    by launches x time are reported: frac = flop_per_launch / avg_launch_us / 2.5 PFLOP/s.  `share` = the group's part of
    the traced launches' summed time.  The same kernel names appear in profiles/r03_train_b32_kernel_stats.csv (in-step
    durations: three streams share the chip there, so they are longer)."""
    K.TRACE = []
    try:
        tr.step(ldr, hdr, gt, update=False)
        torch.cuda.synchronize()
        trace = K.TRACE
    finally:
        K.TRACE = None
    groups = {}
    for e in trace:
        key = (e["kind"], e["kernel"], e["shape"])
        g = groups.setdefault(key, dict(e, labels=[], launches=0))
        g["launches"] += 1
        if e["label"] and e["label"] not in g["labels"]:
            g["labels"].append(e["label"])
    rows = []
    for (kind, kernel, shape), g in groups.items():
        us = _graph_time(torch, g["relaunch"], iters, warm=2)
        rows.append({"layers": g["labels"][:6] + (["+%d more" % (len(g["labels"]) - 6)] if len(g["labels"]) > 6 else []),
                     "kind": kind, "kernel": kernel, "shape": shape, "launches_per_step": g["launches"],
                     "flop_per_launch": g["flop"], "avg_launch_us": round(us, 2),
                     "achieved": round(g["flop"] / (us * 1e-6) / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(g["flop"] / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4), "step_us": round(us * g["launches"], 1)})
    total = sum(r["step_us"] for r in rows) or 1.0
    flops = sum(r["flop_per_launch"] * r["launches_per_step"] for r in rows)
    for r in rows:
        r["share"] = round(r["step_us"] / total, 4)
    rows.sort(key=lambda r: -r["step_us"])
    return {"bound": "mfma", "note": "each row timed alone (back-to-back launches of one hipGraph); step_us = launches_per_step x "
                                     "avg_launch_us; share = of the traced matrix-core launches' summed time",
            "traced_launches": len(trace), "traced_gflop_per_step": round(flops / 1e9, 1), "traced_us_per_step": round(total, 1),
            "traced_frac_of_peak": round(flops / (total * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4), "top": rows[:top], "_all_rows": rows,
            "by_kind_us": {k: round(sum(r["step_us"] for r in rows if r["kind"] == k), 1) for k in sorted({r["kind"] for r in rows})}}


def _precise_instantiation(name):
    """True for the fp32-class (BF16X3, PRECISE = true) instantiations of the matrix-core kernels and for the kernels only
    the fp32-class step launches - a profile that holds them was not a profile of bench-mode steps alone."""
    import re
    m = re.search(r"conv_igemm_kernel<([^>]*)>", name)
    if m:
        a = [t.strip() for t in m.group(1).split(",")]
        return len(a) > 6 and a[6] == "true"
    m = re.search(r"fc_mfma_kernel<([^>]*)>", name)
    if m:
        a = [t.strip() for t in m.group(1).split(",")]
        return len(a) > 1 and a[1] == "true"
    m = re.search(r"da_conv_kernel<([^>]*)>", name)
    if m:
        return m.group(1).split(",")[0].strip() == "true"
    return "conv_wgrad_kernel<" in name      # the register-staged weight gradient: BF16X3 / resize-fused layers only


def in_step_family(roof_top_rows, batch, da, csv_name="r05_train_b32_kernel_stats.csv"):
    """The dominant kernel FAMILIES inside the three-stream step, from a RECORDED profile: FLOP per step = the traced launches
    of the family (the plan, live); time per step = calls x average duration of the family's instantiations in the committed
    rocprofv3 kernel statistics profiles/<csv> of `bench.py --workload train --steps-only` (nothing but bench-mode steps in
    the process), divided by the steps of that profile (= calls of rmsprop2_kernel, one per bench-mode step).  The sidecar
    profiles/<csv>.json records the command, batch and commit of that profile; the object is only emitted for a run of the
    same batch / distortion-aware setting, refused if the statistics hold a fp32-class (PRECISE) instantiation, and is marked
    recorded=true: in-step durations are longer than the alone-on-the-chip ones of roofline_top (three streams share the chip)."""
    path = os.path.join(ROOT, "profiles", csv_name)
    side = path + ".json"
    if not (os.path.exists(path) and os.path.exists(side)):
        return None
    with open(side) as f:
        rec = json.load(f)
    if rec.get("batch") != batch or sorted(rec.get("distortion_aware", [])) != sorted(da):
        return None
    import csv
    with open(path) as f:
        rows = list(csv.DictReader(f))
    if any(_precise_instantiation(r["Name"]) for r in rows):
        return None
    steps = sum(int(r["Calls"]) for r in rows if "rmsprop2_kernel" in r["Name"])
    if steps <= 0:
        return None
    out = {}
    fam_flop = {"conv_igemm_kernel": sum(r["flop_per_launch"] * r["launches_per_step"] for r in roof_top_rows if "conv_igemm_kernel" in r["kernel"]),
                "weight gradients (conv_wgrad*_kernel + wgrad_reduce_kernel)": sum(r["flop_per_launch"] * r["launches_per_step"] for r in roof_top_rows if r["kind"] == "wgrad"),
                "resconv_kernel": sum(r["flop_per_launch"] * r["launches_per_step"] for r in roof_top_rows if r["kind"] == "resconv")}
    match = {"conv_igemm_kernel": ("conv_igemm_kernel",), "weight gradients (conv_wgrad*_kernel + wgrad_reduce_kernel)": ("conv_wgrad", "wgrad_reduce_kernel"),
             "resconv_kernel": ("resconv_kernel",)}
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    total_calls = sum(int(r["Calls"]) for r in rows)
    for fam, keys in match.items():
        ns = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in keys))
        calls = sum(int(r["Calls"]) for r in rows if any(k in r["Name"] for k in keys))
        us = ns / steps / 1e3
        out[fam] = {"gflop_per_step": round(fam_flop[fam] / 1e9, 1), "launches_per_step": round(calls / steps, 1), "us_per_step": round(us, 1),
                    "share_of_kernel_time": round(ns / total_ns, 4), "achieved": round(fam_flop[fam] / (us * 1e-6) / 1e12, 1) if us else None,
                    "frac": round(fam_flop[fam] / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4) if us else None}
    return {"recorded": True, "source": "profiles/%s" % csv_name, "command": rec.get("command"), "commit": rec.get("commit"),
            "steps_bf16": steps, "launches_per_step": round(total_calls / steps, 1),
            "kernel_us_per_step": round(total_ns / steps / 1e3, 1), "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS, "families": out}


def hbm_rooflines(torch, K, batch=32, iters=20):
    """The HBM-bound kernels of the step timed live (their own tensors, hipGraph replay, HIP events): the Dense RMSprop that
    recomputes its gradient (+ bf16 re-pack), the same from a materialised gradient, the fc1 forward weight stream and
    the fc1 weight-gradient write.  achieved = algorithmic
    bytes per launch / average launch time, against the 8 TB/s HBM3E peak."""
    dev = torch.device("cuda", torch.cuda.current_device())
    Kd, N = 8192, 4096
    w = torch.randn(Kd, N, device=dev) * 0.01
    g, ms = torch.randn(Kd, N, device=dev) * 1e-3, torch.zeros(Kd, N, device=dev)
    pf = K.PackedFC(w, precise=False)
    x = torch.randn(batch, Kd, device=dev)
    dy = torch.randn(batch, N, device=dev)
    db = torch.zeros(N, device=dev)
    rows = [
        ("fc_xtdy_kernel<fused> (fc1 8192x4096 RMSprop with its gradient recomputed from M=%d rows: w, ms read; w, ms, two "
         "bf16 images written)" % batch, 20 * Kd * N + 4 * batch * (Kd + N),
         lambda: K.rmsprop_fc_fused(w, ms, x, dy, pf, 1e-4, db=db)),
        ("rmsprop_fc_kernel (the same update from a materialised gradient - data-parallel all-reduce modes: w, g, ms "
         "read; w, ms, two bf16 images written)", 24 * Kd * N, lambda: K.rmsprop_fc(w, g, ms, pf, 1e-4)),
        ("fc_mfma_kernel (fc1 forward, M=%d: bf16 weights streamed once)" % batch, 2 * Kd * N + 4 * batch * (Kd + 4 * N),
         lambda: K.fc_fwd(x, pf, K.BF16)),
        ("fc_xtdy_kernel<store> (fc1 weight gradient, M=%d: fp32 gradient written once)" % batch,
         4 * Kd * N + 4 * batch * (Kd + N), lambda: K.fc_wgrad_bf16(x, dy, g, db)),
    ]
    out = []
    for name, nbytes, fn in rows:
        us = _graph_time(torch, fn, iters, warm=3)
        gbs = nbytes / (us * 1e-6) / 1e9
        out.append({"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "avg_launch_us": round(us, 2), "algorithmic_bytes": nbytes})
    return out


PARITY_FIT_STEPS = 1200    # optimizer steps of train.fit_synthetic behind the parity object (profiles/r03_trained_like.txt:
                           # PSNR(y_gamma, target) 15 dB at random init, 34.7 dB after 500 steps, 35-41 dB up to 3000)


def parity_object(torch, mods, dev, nets_np, batch, oracle_outputs=None):
    """north_star: 'output PSNR within 0.05 dB of the reference'.  At random initialisation that clause is true by
    construction (PSNR(output, target) = 14 dB: any error 48 dB down moves it by 0.002 dB), so the comparison is made at
    TRAINED-LIKE weights from a committed procedure instead of a weight blob: PARITY_FIT_STEPS steps of the product's own
    captured training step on seeded synthetic batches (<pkg>/train.py::fit_synthetic, what `python -m <pkg>.train` runs) in the
    fp32-class BF16X3 mode - the stand-in for a model the REFERENCE trained: weights fitted by the bf16 step are adapted to
    bf16 arithmetic and score 0.03-0.06 dB better in the mode that trained them (HDRSKY_PARITY_FIT=bf16 reproduces that:
    profiles/parity_fit_mode.py, profiles/LABNOTES.md r3 section 0) - then the inference graph on a held-out seeded batch in the bench mode (HDRSKY_BF16)
    and in BF16X3:
      psnr_*_vs_target_db          PSNR of y_final_gamma against hdr_logCompression(hdr_t), whole batch
      psnr_bf16_vs_x3_db, q_max_db the two modes against each other; q_max = that - 19.4 dB is the output quality up to
                                   which an independent error of this size stays below 0.05 dB (10 log10(1 + 10^-1.94))
      *_vs_oracle_db, delta_psnr_vs_oracle_target_db (when `oracle_outputs` is given: the CPU restatement's fp32
                                   y_final_gamma of the first images, computed by the cpu_baseline leg - the checker, not the
                                   product): both modes against it, and PSNR(bf16, target) - PSNR(oracle, target)."""
    params, synth, engine, trainer, K, train = (mods[m] for m in ("params", "synth", "engine", "trainer", "kernels", "train"))
    gen, sun, dis, vgg = nets_np
    x3fit = os.environ.get("HDRSKY_PARITY_FIT", "x3") != "bf16"
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=x3fit, compute=K.BF16X3 if x3fit else K.BF16)
    t0 = time.perf_counter()
    train.fit_synthetic(tr, PARITY_FIT_STEPS, batch, seed0=0)
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    held = synth.make_batch_device(batch, seed=999_999, device=dev)
    gen_t = {k[4:]: v.detach().clone() for k, v in tr.gs.w.items() if k.startswith("gen.")}
    sun_t = {k[4:]: v.detach().clone() for k, v in tr.gs.w.items() if k.startswith("sun.")}
    del tr
    torch.cuda.empty_cache()
    nets = engine.Nets(gen_t, sun_t, device=dev, precise=True)
    y16 = engine.generator_forward(nets, held["ldr"], compute=K.BF16)["y_final_gamma"]
    y3 = engine.generator_forward(nets, held["ldr"], compute=K.BF16X3)["y_final_gamma"]
    tgt = K.tonemap(held["hdr_t"], False)
    torch.cuda.synchronize()
    peak = float(tgt.abs().max())
    p16, p3 = train.psnr_db(y16, tgt, peak), train.psnr_db(y3, tgt, peak)
    pm = train.psnr_db(y16, y3, float(y3.abs().max()))
    out = {"weights": "trained-like: %d steps of train.fit_synthetic in %s at batch %d (seeded device-side synthetic batches, %.1f s), "
                      "then a held-out seeded batch" % (PARITY_FIT_STEPS, "BF16X3" if x3fit else "HDRSKY_BF16", batch, fit_s),
           "images": int(batch), "psnr_bf16_vs_target_db": round(p16, 4), "psnr_x3_vs_target_db": round(p3, 4),
           "delta_psnr_vs_target_db": round(p16 - p3, 4), "psnr_bf16_vs_x3_db": round(pm, 2), "q_max_db": round(pm - 19.4, 2),
           "within_0p05_db": bool(abs(p16 - p3) <= 0.05)}
    if oracle_outputs is not None:
        n, yo = oracle_outputs({k: v.cpu().numpy() for k, v in gen_t.items()}, {k: v.cpu().numpy() for k, v in sun_t.items()},
                               held["ldr"].cpu().numpy())
        yo = torch.from_numpy(yo).to(dev)
        pk = float(yo.abs().max())
        sub_peak = float(tgt[:n].abs().max())
        # the graph couples the images of a batch (tf.reduce_max over the batch tensor, generator.py:160): the subset goes
        # through the GPU graph as a batch of its own, like through the oracle
        sub = held["ldr"][:n].contiguous()
        y16n = engine.generator_forward(nets, sub, compute=K.BF16)["y_final_gamma"]
        y3n = engine.generator_forward(nets, sub, compute=K.BF16X3)["y_final_gamma"]
        po, p16n, p3n = (train.psnr_db(y, tgt[:n], sub_peak) for y in (yo, y16n, y3n))
        out.update({"oracle_images": n, "psnr_bf16_vs_oracle_db": round(train.psnr_db(y16n, yo, pk), 2),
                    "psnr_x3_vs_oracle_db": round(train.psnr_db(y3n, yo, pk), 2),
                    "psnr_oracle_vs_target_db": round(po, 4),
                    "delta_psnr_vs_oracle_target_db": round(p16n - po, 4), "delta_psnr_x3_vs_oracle_target_db": round(p3n - po, 4),
                    "within_0p05_db": bool(abs(p16 - p3) <= 0.05 and abs(p16n - po) <= 0.05)})
    return out


def oracle_outputs_fn(torch, n_images=8):
    """Part of the cpu_baseline leg (the only place bench.py touches oracle/): returns f(gen, sun, ldr) -> (n, fp32
    y_final_gamma of the first n images through oracle/step.inference) for the parity object - the oracle as the checker."""
    def f(gen_np, sun_np, ldr_np):
        from oracle import step as ostep
        torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
        tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
        out = ostep.inference(tt(gen_np), tt(sun_np), torch.from_numpy(ldr_np[:n_images]))
        return n_images, out["y_final_gamma"].numpy()
    return f


CPU_THREADS = 16       # the GPU box's CPU share per GPU.  Measured there (profiles/cpu_threads_probe.py, 256 logical cores
                       # shared with other tenants): the batch-32 training step of the restatement takes 0.56 s on 16 threads,
                       # 0.84 s on 32, 1.8 s on 64 and 16 s on torch's default of 128 (oversubscription) - round 2's 6.8
                       # images/s was that default, not what the host can do


def cpu_baseline(torch, workload, nets_np, batch_np, budget_s=24.0, iters=20, warmups=3):
    """CPU restatement (oracle/, NOT TensorFlow) of the same workload on this host's cores, BASELINE.md's protocol (3
    warm-ups, median of >= 20 iterations) on CPU_THREADS threads; a BOUNDED sample: the sample batch is halved from the
    bench batch only if warm-ups + iterations would not fit the time budget."""
    from oracle import step as ostep
    torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
    tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    gen, sun, dis, vgg = (tt(d) for d in nets_np)
    ldr, hdr, gt = (torch.from_numpy(batch_np[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    if workload == "fwd":
        fn = lambda n: ostep.inference(gen, sun, ldr[:n])
        what = "oracle/step.inference"
    else:
        fn = lambda n: ostep.train_step_grads(gen, sun, dis, vgg, ldr[:n], hdr[:n], gt[:n])
        what = "oracle/step.train_step_grads (forward + both backward passes; optimizer excluded)"
    n = int(ldr.shape[0])
    t_start = time.perf_counter()
    while True:                     # first warm-up doubles as the probe that sizes the sample
        t0 = time.perf_counter(); fn(n); t1 = time.perf_counter() - t0
        if n == 1 or t1 * (iters + warmups - 1) <= budget_s:
            break
        n = max(1, n // 2)
    for _ in range(warmups - 1):
        fn(n)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter(); fn(n); times.append(time.perf_counter() - t0)
    times.sort()
    med = 0.5 * (times[(iters - 1) // 2] + times[iters // 2])
    return {"value": round(n / med, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "median of %d iterations after %d warm-ups, batches of %d images of the same synthetic workload through "
                      "%s (torch-CPU fp32 restatement, not TF2); %.1f s in all" % (iters, warmups, n, what,
                                                                                   time.perf_counter() - t_start)}


def fp32_class_step(torch, dist, trainer, K, nets_np, data, batch, dev, da, steps=10, warmup=3):
    """The same captured training step in the mode that meets the fp32 tolerance against the oracle (HDRSKY_BF16X3: three
    bf16 MFMA products of hi / lo split fp32 operands, fp32 accumulation; the reference's arithmetic is fp32, ops.py:41-42):
    what the fp32-class result costs beside the bf16 line `value` is quoted on.  Rank 0, one GPU, after the main timing."""
    gen, sun, dis, vgg = nets_np
    tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=True, compute=K.BF16X3, world_size=1, distortion_aware=da)
    out = tr.capture(*data)
    dt = timed(torch, dist, lambda: tr.replay(), steps, warmup, False, dev)
    assert torch.isfinite(out["y_final_lin"]).all() and torch.isfinite(tr.gs.flat).all()
    res = {"mode": "BF16X3 (hi*hi + lo*hi + hi*lo, fp32 accumulate): per-operator 2e-4, losses 2e-3 against the fp32 oracle",
           "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4), "images_per_s": round(batch * steps / dt, 1),
           "algorithmic_tflops": round(batch * steps / dt * TRAIN_MFLOP_PER_IMG * 1e6 / 1e12, 2)}
    del tr, out
    torch.cuda.empty_cache()
    return res


def _no_gc():
    """kernels.no_gc: no cyclic-garbage finaliser (an old hipGraphExec) in the middle of a hipGraph capture."""
    return importlib.import_module(PKG + ".kernels").no_gc()


def timed(torch, dist, one_step, steps, warmup, dp, dev, finish=None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks (seconds).
    finish (the trainer's flush): work a step may leave pending for the next one - the deferred Dense update of
    Trainer(defer_dense=True), which opens the NEXT replay - is completed INSIDE the timed region, after the K-th step and after
    the warm-up: the region holds exactly K whole steps' work."""
    for _ in range(warmup):
        one_step()
    if finish is not None:
        finish()
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    if finish is not None:
        finish()
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dp:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt


def capture_forward(torch, fn, dp, no_graph):
    """Warm-up on a side stream (lazy kernel attributes, allocator), then the whole pass as one hipGraph."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            out = fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if no_graph:
        return fn, out
    g = torch.cuda.CUDAGraph()
    # thread_local: RCCL's watchdog thread may query events while this thread captures
    with _no_gc(), torch.cuda.graph(g, capture_error_mode="thread_local" if dp else "global"):
        out = fn()
    return g.replay, out


def hires_workload(torch, mods, dev, batch):
    """configs[4] on one GPU (see the module docstring).  Returns (step function, probe tensor getter, sun-pose FC row)."""
    params, engine, K = mods["params"], mods["engine"], mods["kernels"]
    disc_mod, vgg_mod = mods["discriminator"], mods["vgg16"]
    H, W = 128, 512
    gen = params.init_params(params.generator_spec(H, W), 0)
    nets = engine.Nets(gen, None, device=dev, precise=False, im_height=H, im_width=W)
    g = torch.Generator(device=dev); g.manual_seed(1234)
    ldr = torch.rand(batch, H, W, 3, device=dev, generator=g)
    hdr = torch.rand(batch, H, W, 3, device=dev, generator=g) * 4.0
    rad = torch.rand(batch, H, W, 3, device=dev, generator=g)
    dis = disc_mod.model(device=dev, compute=K.BF16)
    vgg = vgg_mod.Vgg16(weights=params.init_params(params.vgg_spec(), 3), device=dev, compute=K.BF16)
    losses = torch.zeros(4, device=dev)
    state = {}

    def step():
        losses.zero_()
        res = engine.encode(nets, ldr, K.BF16)
        res_da = engine.encode(nets, ldr, K.BF16, distortion_aware=True)
        sky = engine.decode(nets, res, "f", ldr, K.BF16)
        sunp = engine.decode(nets, res_da, "u", rad, K.BF16)
        y_gamma, y_lin, _, _, _ = K.blend(sky, sunp, engine.THRESHOLD, extras=False)
        d_fake = dis([ldr, y_lin], training=False)
        K.mse(d_fake, 1.0, 1.0, 1.0, losses[0:1], want_grad=False)                 # LSGAN (train.py:327)
        K.l1(y_lin, hdr, 1.0, 0.0, losses[1:2])                                   # L1 (train.py:324)
        scratch = torch.empty_like(y_lin)
        K.dog_loss(y_lin, hdr, 1.0, losses[2:3], scratch)                         # DoG pyramid (train.py:316-322)
        for p, q in zip(vgg(y_gamma), vgg(K.tonemap(hdr, False))):                # perceptual (train.py:308-313)
            K.l1(p, q, 1.0, 0.0, losses[3:4])
        state["y"] = y_lin
        return state

    # the sun-pose net's first Dense layer at 128x512 as a weight-streaming GEMM slice: K = 16*64*128 inputs, 4096 of
    # its 65536 outputs (1.07 GB of bf16 weights; the full layer is 16 such slices per GPU, or one per GPU when sharded)
    Kd, N = (H // 8) * (W // 8) * 128, 4096
    wfc = torch.randn(Kd, N, device=dev) * 0.01
    pf = K.PackedFC(wfc, precise=False, need_dgrad=False)
    del wfc
    xfc = torch.randn(batch, Kd, device=dev)
    us = _graph_time(torch, lambda: K.fc_fwd(xfc, pf, K.BF16), 10, warm=2)
    nbytes = 2 * Kd * N
    fc_row = {"bound": "hbm", "kernel": "fc_mfma_kernel (128x512 sun-pose fc1 slice %dx%d, M=%d)" % (Kd, N, batch),
              "achieved": round(nbytes / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_us": round(us, 1),
              "algorithmic_bytes": nbytes, "slices_in_full_layer": 16}
    return step, (lambda: state["y"]), fc_row


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside a torchrun job: this process - which has not touched the GPU and never
    will - starts the N ranks as a CHILD `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>`
    (never an exec: a process that initialised the GPU must not be replaced), relays what rank 0 prints and returns the
    child's exit code (non-zero as soon as any rank fails).  Fewer than N devices: refuse instead of timing fewer GPUs."""
    import subprocess
    import torch                    # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    one_card = os.environ.get("HDRSKY_BENCH_ONE_CARD", "0") == "1"      # rehearsal: every rank on cuda:0 over gloo
    if have < args.gpus and not one_card:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible - refusing to report an n_gpus=%d number from "
                         "fewer devices\n" % (args.gpus, have, args.gpus))
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:          # a torchrun job of another size than the flag says: never report the wrong n_gpus
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # HDRSKY_BENCH_FORCE_DP=1 runs the N>1 code path (process group, phase split, overlapped all-reduces) with a
    # single rank, so that path can be rehearsed on a one-GPU box.
    dp = world > 1 or os.environ.get("HDRSKY_BENCH_FORCE_DP", "0") == "1"
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # Rehearsal of the N>1 path on a one-GPU box: HDRSKY_BENCH_ONE_CARD=1 puts every rank on cuda:0 and
        # HDRSKY_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).  Never set by the driver.
        if os.environ.get("HDRSKY_BENCH_ONE_CARD", "0") == "1":
            local = 0
        backend = os.environ.get("HDRSKY_DIST_BACKEND", "nccl")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]            # the image exports VERSION: RCCL would print its banner to STDOUT, in
                                                    # front of the one JSON line this script owes its caller
        torch.cuda.set_device(local)
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))
        comm_ranks, comm_backend = dist.get_world_size(), ("rccl" if backend == "nccl" else backend)
    else:
        comm_ranks, comm_backend = 1, None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if dp else 0)

    mods = {m: importlib.import_module(PKG + "." + m) for m in
            ("params", "synth", "engine", "trainer", "parallel", "kernels", "discriminator", "vgg16", "train")}
    params, synth, engine, trainer, par, K = (mods[m] for m in ("params", "synth", "engine", "trainer", "parallel", "kernels"))
    hires = args.workload == "hires"
    hires_train = args.workload == "hires-train"
    batch = args.batch if args.batch is not None else (8 if (hires or hires_train) else 32)

    gen = params.init_params(params.generator_spec(), 0)
    if args.roofline_only:
        w = torch.from_numpy(gen["res.0.conv1.w"]).to(dev)
        print(json.dumps(dominant_kernel_roofline(torch, K, K.PackedConv(w, False), batch, 32, 128)))
        return
    res = {"n_gpus": world, "rccl_ranks": comm_ranks if comm_backend in ("rccl", None) else 0, "comm_backend": comm_backend,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16"}
    fc_row = None
    if hires_train:
        H, W = 128, 512
        gen = params.init_params(params.generator_spec(H, W), 0)
        dis = params.init_params(params.discriminator_spec(), 2)
        vgg = params.init_params(params.vgg_spec(), 3)
        g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
        ldr = torch.round(torch.rand(batch, H, W, 3, device=dev, generator=g) * 255.0) / 255.0
        hdr = ldr ** 2.2 * (1.0 + 3.0 * torch.rand(batch, H, W, 3, device=dev, generator=g))
        cmf = torch.softmax(3.0 * torch.randn(batch, H * W, device=dev, generator=g), dim=1).contiguous()
        gt = torch.softmax(3.0 * torch.randn(batch, H * W, device=dev, generator=g), dim=1).contiguous()
        cams = [torch.relu(torch.randn(batch, H >> i, W >> i, 1, device=dev, generator=g)).contiguous() for i in range(3)]
        tr = trainer.Trainer(gen, None, dis, vgg, device=dev, precise=False, compute=K.BF16, world_size=world, im_height=H,
                             im_width=W, distortion_aware=args.da, sunpose="external")
        par.broadcast_params_([tr.gs.flat, tr.ds.flat])
        tr.repack()
        ex = par.GradientExchange(tr, device=dev, mode=args.dp_mode)
        hooks, pre_hooks = (ex.hooks, ex.pre_hooks) if dp else (None, None)
        if args.no_graph:
            out = tr.step(ldr, hdr, gt, update=False, cmf=cmf, cams=cams)
            one_step = lambda: (tr.step(ldr, hdr, gt, update=False, cmf=cmf, cams=cams), dp and ex.reduce_all(), tr.apply_gradients())
        else:
            out = tr.capture(ldr, hdr, gt, cmf=cmf, cams=cams)
            one_step = lambda: tr.replay(hooks=hooks, pre_hooks=pre_hooks)
        dt = timed(torch, dist, one_step, args.steps, args.warmup, dp, dev)
        assert torch.isfinite(out["y_final_lin"]).all() and torch.isfinite(tr.gs.flat).all()
        imgs = batch * world * args.steps
        res.update({
            "metric": "training images/sec (128x512 sky panoramas)", "value": round(imgs / dt, 1), "unit": "images/s",
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_img": round(dt / imgs * world * 1e3, 6),
            "data": "synthetic (uniform random 128x512 panoramas, random-init weights, synthetic VGG16 weights, random "
                    "sun-position map and Grad-CAM maps)",
            "config": {"workload": "BASELINE configs[4] on %d GPU(s): full train.py step at 128x512, batch=%d per GPU - generator "
                                   "+ sun-radiance head, discriminator x3, VGG16 perceptual, DoG/L1/LSGAN losses, both "
                                   "backward passes, RMSprop x2.  SUBSTITUTION (SURVEY.md section 8d): the 12.9 G-parameter "
                                   "sun-pose net is replaced by its outputs (cmf + three Grad-CAM maps are inputs of the "
                                   "step, Trainer(sunpose='external')); the KL term is a constant of such a step" % (world, batch),
                       "per_gpu_batch": batch, "global_batch": batch * world,
                       "parallelism": ("dp%d over %s, %d ranks (conv + discriminator slices all-reduced)" % (world, comm_backend, comm_ranks))
                                      if world > 1 else "single",
                       "hipgraph": not args.no_graph, "distortion_aware": sorted(engine.da_parts(args.da))},
            "algorithmic_tflops": round(imgs / dt * HIRES_TRAIN_MFLOP_PER_IMG * 1e6 / 1e12, 2)})
        del tr, ex, one_step, out
    elif hires:
        step, probe, fc_row = hires_workload(torch, mods, dev, batch)
        one_step, out = capture_forward(torch, step, dp, args.no_graph)
        dt = timed(torch, dist, one_step, args.steps, args.warmup, dp, dev)
        assert torch.isfinite(probe()).all()
        imgs = batch * world * args.steps
        res.update({
            "metric": "forward + losses images/sec (128x512 sky panoramas)", "value": round(imgs / dt, 1), "unit": "images/s",
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_img": round(dt / imgs * world * 1e3, 6),
            "data": "synthetic (uniform random 128x512 panoramas, random-init weights, synthetic VGG16 weights)",
            "config": {"workload": "BASELINE configs[4] on one GPU: 128x512, batch=%d per GPU - generator encoder with plain "
                                   "and with distortion-aware res blocks, both decoders, blending, discriminator, VGG16 x2, "
                                   "L1 / DoG / perceptual / LSGAN loss values (forward + losses).  SUBSTITUTION (SURVEY.md "
                                   "section 8d): the 12.9 G-parameter sun-pose net is not in the step; its fc1 is timed "
                                   "separately as a weight-streaming GEMM slice (sunpose_fc)" % batch,
                       "per_gpu_batch": batch, "global_batch": batch * world,
                       "parallelism": ("replicas x%d (%s barrier only)" % (world, comm_backend)) if world > 1 else "single",
                       "hipgraph": not args.no_graph},
            "algorithmic_tflops": round(imgs / dt * HIRES_MFLOP_PER_IMG * 1e6 / 1e12, 2)})
    else:
        sun = params.init_params(params.sunpose_spec(), 1)
        dis = params.init_params(params.discriminator_spec(), 2)
        vgg = params.init_params(params.vgg_spec(), 3)
        batch_np = synth.make_batch(batch, seed=1234 + rank)
        ldr = torch.from_numpy(batch_np["ldr"]).to(dev)
        hdr = torch.from_numpy(batch_np["hdr_t"]).to(dev)
        gt = torch.from_numpy(batch_np["sunpose_gt"]).to(dev)
        res["data"] = "synthetic (seeded sky-dome + sun lobe + sensor noise, random-init weights, synthetic VGG16 weights)"
        do_train, do_fwd = args.workload in ("all", "train"), args.workload in ("all", "fwd")
        roof_pw, roof_top = None, None
        if do_train:
            # --defer-dense: the Dense kernels' RMSprop launch (1 GB of HBM traffic that nothing in the step waits for) opens the NEXT
            # replay, in the window one stream idles in beside the forward pass, instead of closing the step (trainer.Trainer);
            # tr.flush() inside the timed region applies the last one: K timed steps hold K whole updates.  Measured level with the
            # default plan (2.566 vs 2.566 ms): the forward pass it runs beside slows down by what the tail of the step wins.
            tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16, world_size=world,
                                 distortion_aware=args.da, defer_dense=args.defer_dense)
            # One step = the Trainer's segment plan (forward, losses, both backward passes, RMSprop x2 + weight
            # re-packing), every segment captured into its own hipGraph and replayed on its stream.  N > 1: each replica
            # runs the reference's batch-32 step on its shard; gradients are summed over replicas (RCCL; every loss is a
            # batch mean, so the data-parallel gradient is the replica average: SURVEY.md section 8e) and scaled by
            # 1/world inside the RMSprop kernel.  How the sum travels is parallel.GradientExchange's mode.
            par.broadcast_params_([tr.gs.flat, tr.ds.flat])   # replicas start from rank 0's weights
            tr.repack()
            ex = par.GradientExchange(tr, device=dev, mode=args.dp_mode)   # hooks on the segment plan: see parallel.py
            hooks, pre_hooks = (ex.hooks, ex.pre_hooks) if dp else (None, None)
            roof_pw = tr.conv["gen.res.0.conv1"].pk
            if args.no_graph:
                out = tr.step(ldr, hdr, gt, update=False)
                one_step = lambda: (tr.step(ldr, hdr, gt, update=False), dp and ex.reduce_all(), tr.apply_gradients())
            else:
                out = tr.capture(ldr, hdr, gt)
                one_step = lambda: tr.replay(hooks=hooks, pre_hooks=pre_hooks)
            dt = timed(torch, dist, one_step, args.steps, args.warmup, dp, dev, finish=tr.flush)
            assert torch.isfinite(out["y_final_lin"]).all()
            if rank == 0 and not args.no_roofline_top:
                roof_top = roofline_top(torch, K, tr, ldr, hdr, gt, top=args.roofline_rows or 10 ** 6)
            imgs = batch * world * args.steps
            res.update({
                "metric": "training images/sec (32x128 sky panoramas); generator fwd ms/img in `fwd`" if do_fwd else
                          "training images/sec (32x128 sky panoramas)",
                "value": round(imgs / dt, 1), "unit": "images/s",
                "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_img": round(dt / imgs * world * 1e3, 6),
                "config": {"workload": "BASELINE configs[2]: full train.py step (gen + sunpose + disc + VGG16 perceptual + "
                                       "tone-map/DoG/L1/KL/LSGAN losses, RMSprop x2), batch=%d per GPU, 32x128x3" % batch,
                           "per_gpu_batch": batch, "global_batch": batch * world,
                           "parallelism": ("dp%d over %s, %d ranks (%s: %s)" % (world, comm_backend, comm_ranks, ex.mode, ex.describe()))
                                          if world > 1 else "single",
                           "hipgraph": not args.no_graph, "distortion_aware": sorted(engine.da_parts(args.da)),
                           "dense_update": "deferred into the next replay's forward pass (flushed inside the timed region)" if tr._defer else "inside its step"},
                "algorithmic_tflops": round(imgs / dt * TRAIN_MFLOP_PER_IMG * 1e6 / 1e12, 2)})
            del tr, ex, one_step, out
            torch.cuda.empty_cache()
        if do_fwd:
            nets = engine.Nets(gen, sun, device=dev, precise=False)
            roof_pw = roof_pw if roof_pw is not None else nets.pk["gen.res.0.conv1"]
            # ONE hipGraph with the fork / join of the two branches inside.  Round 5 measured the alternative - one graph per branch
            # on its own stream (engine.ForwardGraphs, profiles/fwd_branches.py -> profiles/r05_fwd_branches.txt): 0.571 ms against
            # 0.509 ms.  The runtime does start the encoder branch of the single graph late (profiles/r04_fwd_timeline.txt), but
            # that is no loss: side by side the two branches slow each other down by more than they overlap (sun branch 373 us
            # alone, 499 us beside the encoder branch; encoder branch 237 -> 388 us) - the launches share the chip, they do not
            # wait for it.
            one_step, out = capture_forward(torch, lambda: engine.generator_forward(nets, ldr, compute=K.BF16, distortion_aware=args.da), dp, args.no_graph)
            dtf = timed(torch, dist, one_step, args.steps, args.warmup, dp, dev)
            assert torch.isfinite(out["y_final_lin"]).all()
            imgs = batch * world * args.steps
            fwd = {"ms_per_step": round(dtf / args.steps * 1e3, 4), "ms_per_img": round(dtf / imgs * world * 1e3, 6),
                   "images_per_s": round(imgs / dtf, 1),
                   "algorithmic_tflops": round(imgs / dtf * FWD_MFLOP_PER_IMG * 1e6 / 1e12, 2),
                   "workload": "BASELINE configs[1]: generator + sunpose_net forward (incl. the Grad-CAM sweep and "
                               "sun-radiance head of the generator graph), batch=%d per GPU, 32x128x3%s" %
                               (batch, ", replicas" if world > 1 else "")}
            if do_train:
                res["fwd"] = fwd
            else:
                res.update({"metric": "generator fwd images/sec (32x128 sky panoramas)", "value": fwd["images_per_s"],
                            "unit": "images/s", "ms_per_step": fwd["ms_per_step"], "ms_per_img": fwd["ms_per_img"],
                            "config": {"workload": fwd["workload"], "per_gpu_batch": batch, "global_batch": batch * world,
                                       "parallelism": ("replicas x%d (%s barrier only)" % (world, comm_backend)) if world > 1 else "single",
                                       "hipgraph": not args.no_graph},
                            "algorithmic_tflops": fwd["algorithmic_tflops"]})
            del one_step, out
    # the single-GPU probes run without a process group (RCCL's communicator closed, the other ranks gone)
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if hires_train:
            if args.da:       # the best-utilised launch of a distortion-aware 128x512 step (the driver line's roofline object is resconv's)
                res["roofline"] = gemm1x1_roofline(torch, K)
        elif hires:
            res["roofline_hbm"] = [fc_row]
            res["sunpose_fc"] = fc_row
        elif args.steps_only:
            pass
        else:
            res["roofline"] = dominant_kernel_roofline(torch, K, roof_pw, batch, 32, 128)
            if roof_top is not None:
                res["roofline_top"] = roof_top
                ins = in_step_family(roof_top.pop("_all_rows"), batch, sorted(engine.da_parts(args.da)))
                if ins is not None:
                    res["roofline"]["in_step"] = ins
            res["roofline_hbm"] = hbm_rooflines(torch, K, batch)
            if do_train and not args.no_graph:
                res["fp32_class"] = fp32_class_step(torch, dist, trainer, K, (gen, sun, dis, vgg), (ldr, hdr, gt), batch, dev, args.da)
            if not args.no_parity:
                res["parity"] = parity_object(torch, mods, dev, (gen, sun, dis, vgg), batch,
                                              None if args.no_cpu_baseline else oracle_outputs_fn(torch))
            if not args.no_cpu_baseline:
                nets_np = (gen, sun, dis, vgg)
                if do_train:
                    res["cpu_baseline"] = cpu_baseline(torch, "train", nets_np, batch_np)
                    if do_fwd:
                        res["cpu_baseline_fwd"] = cpu_baseline(torch, "fwd", nets_np, batch_np, budget_s=8.0)
                else:
                    res["cpu_baseline"] = cpu_baseline(torch, "fwd", nets_np, batch_np)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
