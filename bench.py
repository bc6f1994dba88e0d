#!/usr/bin/env python3
"""Throughput bench of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N=1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the workload over one batch of 32 synthetic 32x128 sky panoramas per GPU
(inputs resident in HBM before the timed region; the pass is replayed as hipGraphs).
Rank 0 prints ONE JSON line (metric/value/... + "roofline" + "cpu_baseline").

Workloads (BASELINE.json configs):
  train (default) = configs[2]/[3]: the full train.py step - generator + sun-pose + Grad-CAM + sun-radiance forward,
                    discriminator x3, VGG16 perceptual, DoG/L1/KL/LSGAN losses, both backward passes, RMSprop x2,
                    weight re-packing - batch 32 per GPU; N > 1: data parallel, gradients all-reduced (RCCL).
  fwd             = configs[1]: generator + sun-pose net (+ Grad-CAM sweep) forward only, batch 32 (replicas).
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
FWD_MFLOP_PER_IMG = 3220.3   # G + S + C (algorithmic 2*MAC of conv/dense contractions)
TRAIN_MFLOP_PER_IMG = 16900.0  # 3(G+S) + C + 8D + 3V


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1..3]: 32)")
    ap.add_argument("--workload", default="train", choices=["train", "fwd"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the timed loop of the roofline kernel and print its object (the command behind "
                         "profiles/r01_roofline_kernel_stats.csv: in the full run the same kernel template also "
                         "serves other layers, beside other streams, so its rocprof average there is not this launch)")
    return ap.parse_args()


def dominant_kernel_roofline(torch, K, pw, bias, batch, h, w, iters=200):
    """The res-block convolution (3x3, 128->128 on [B,8,32,128]: 12 forward launches + their data-gradient twins per
    step) timed live with HIP events on the launch stream.  Algorithmic FLOPs per launch = 2*(B*8*32)*(3*3*128)*128."""
    dev = bias.device
    x = torch.randn(batch, h // 4, w // 4, 128, device=dev)
    y = torch.empty_like(x)
    for _ in range(10):
        K.conv2d(x, pw, bias, want_stats=True, compute=K.BF16, out=y)
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    # thread_local: with a process group alive RCCL's watchdog thread queries events while this thread captures
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(iters):
            K.conv2d(x, pw, bias, want_stats=True, compute=K.BF16, out=y)
    g.replay()
    torch.cuda.synchronize()
    e0.record(stream)
    g.replay()
    e1.record(stream)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flop = 2.0 * (batch * (h // 4) * (w // 4)) * (9 * 128) * 128
    achieved = flop / (us * 1e-6) / 1e12
    traffic = None   # HBM-side bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE)
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_resconv.json")
    if batch == 32 and os.path.exists(pmc):
        with open(pmc) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    return {"bound": "mfma", "kernel": "conv_igemm_kernel (res-block 3x3 128->128, B=%d)" % batch,
            "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 5), "traffic": traffic,
            "avg_launch_us": round(us, 3), "flop_per_launch": flop}


def cpu_baseline(torch, workload, nets_np, batch_np):
    """CPU restatement (oracle/, NOT TensorFlow) of the same workload on this host's cores: bounded sample."""
    from oracle import step as ostep
    tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    gen, sun, dis, vgg = (tt(d) for d in nets_np)
    ldr, hdr, gt = (torch.from_numpy(batch_np[k]) for k in ("ldr", "hdr_t", "sunpose_gt"))
    if workload == "fwd":
        fn = lambda n: ostep.inference(gen, sun, ldr[:n])
        what = "oracle/step.inference"
    else:
        fn = lambda n: ostep.train_step_grads(gen, sun, dis, vgg, ldr[:n], hdr[:n], gt[:n])
        what = "oracle/step.train_step_grads (forward + both backward passes; optimizer excluded)"
    fn(2)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 12.0:
        fn(ldr.shape[0])
        n += ldr.shape[0]
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d images (batches of %d) of the same synthetic workload through %s "
                      "(torch-CPU fp32 restatement, not TF2), %.1f s" % (n, ldr.shape[0], what, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
        torch.cuda.set_device(local)
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if dp else 0)

    params = importlib.import_module(PKG + ".params")
    synth = importlib.import_module(PKG + ".synth")
    engine = importlib.import_module(PKG + ".engine")
    trainer = importlib.import_module(PKG + ".trainer")
    par = importlib.import_module(PKG + ".parallel")
    K = importlib.import_module(PKG + ".kernels")

    gen = params.init_params(params.generator_spec(), 0)
    sun = params.init_params(params.sunpose_spec(), 1)
    dis = params.init_params(params.discriminator_spec(), 2)
    vgg = params.init_params(params.vgg_spec(), 3)
    batch_np = synth.make_batch(args.batch, seed=1234 + rank)
    ldr = torch.from_numpy(batch_np["ldr"]).to(dev)
    hdr = torch.from_numpy(batch_np["hdr_t"]).to(dev)
    gt = torch.from_numpy(batch_np["sunpose_gt"]).to(dev)

    if args.roofline_only:
        w = torch.from_numpy(gen["res.0.conv1.w"]).to(dev)
        print(json.dumps(dominant_kernel_roofline(torch, K, K.PackedConv(w, False), torch.zeros(128, device=dev),
                                                  args.batch, 32, 128)))
        return
    if args.workload == "fwd":
        nets = engine.Nets(gen, sun, device=dev, precise=False)
        phases = [lambda: engine.generator_forward(nets, ldr, compute=K.BF16)]
        roof_pw, roof_b = nets.pk["gen.res.0.conv1"], nets.gen["res.0.conv1.b"]
        probe = lambda out: out["y_final_lin"]
    else:
        tr = trainer.Trainer(gen, sun, dis, vgg, device=dev, precise=False, compute=K.BF16, world_size=world)
        # One step = the Trainer's segment plan (forward, losses, both backward passes, RMSprop x2 + weight re-packing),
        # every segment captured into its own hipGraph and replayed on its stream.  N > 1: each replica runs the
        # reference's batch-32 step on its shard; gradients are summed over replicas (RCCL all-reduce; every loss is a
        # batch mean, so the data-parallel gradient is the replica average: SURVEY.md section 8e) and scaled by 1/world
        # inside the RMSprop kernel.  The all-reduce of the sun-pose Dense gradients (201 of the 233 MB) is started as
        # soon as they are complete and runs on RCCL's stream beside the rest of the backward pass; the remaining
        # 32 MB follow when every gradient is ready.
        par.broadcast_params_([tr.gs.flat, tr.ds.flat])   # replicas start from rank 0's weights
        tr.repack()
        ex = par.GradientExchange(tr, device=dev)   # hooks on the segment plan: see parallel.py
        hooks, pre_hooks = (ex.hooks, ex.pre_hooks) if dp else (None, None)
        roof_pw, roof_b = tr.conv["gen.res.0.conv1"].pk, tr.gs.w["gen.res.0.conv1.b"]
        probe = lambda out: out["y_final_lin"]
        if args.no_graph:
            out = tr.step(ldr, hdr, gt, update=False)
            one_step = lambda: (tr.step(ldr, hdr, gt, update=False), dp and [ex.fc_grads_reduce(), ex.grads_ready()],
                                tr.apply_gradients())
        else:
            out = tr.capture(ldr, hdr, gt)
            one_step = lambda: tr.replay(hooks=hooks, pre_hooks=pre_hooks)
        phases = None

    if phases is not None:   # forward workload: warm-up (eager), then the whole forward captured as one hipGraph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                out = phases[0]()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if args.no_graph:
            one_step = phases[0]
        else:
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(g, capture_error_mode="thread_local" if dp else "global"):
                out = phases[0]()
            one_step = g.replay

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dp:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(probe(out)).all()

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / dt
        train = args.workload == "train"
        res = {
            "metric": "training images/sec (32x128 sky panoramas)" if train else "generator fwd images/sec (32x128 sky panoramas)",
            "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_img": round(dt / imgs * world * 1e3, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (seeded sky-dome + sun lobe + sensor noise, random-init weights, synthetic VGG16 weights)",
            "config": {"workload": ("BASELINE configs[2]: full train.py step (gen + sunpose + disc + VGG16 perceptual + "
                                    "tone-map/DoG/L1/KL/LSGAN losses, RMSprop x2), batch=%d per GPU, 32x128x3" % args.batch)
                       if train else
                       ("BASELINE configs[1]: generator + sunpose_net forward (incl. the Grad-CAM sweep and sun-radiance "
                        "head of the generator graph), batch=%d per GPU, 32x128x3" % args.batch),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": ("dp%d (RCCL all-reduce of fp32 gradients)" % world if train else "replicas") if world > 1 else "single",
                       "hipgraph": not args.no_graph},
            "algorithmic_tflops": round(value * (TRAIN_MFLOP_PER_IMG if train else FWD_MFLOP_PER_IMG) * 1e6 / 1e12, 2),
        }
        res["roofline"] = dominant_kernel_roofline(torch, K, roof_pw, roof_b, args.batch, 32, 128)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(torch, args.workload, (gen, sun, dis, vgg), batch_np)
        print(json.dumps(res))
    if dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
