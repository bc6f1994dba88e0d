#!/usr/bin/env python3
"""Throughput bench of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N=1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the workload over one batch of 32 synthetic 32x128 sky panoramas per GPU
(inputs resident in HBM before the timed region; the whole pass is replayed as one hipGraph).
Rank 0 prints ONE JSON line (metric/value/... + "roofline" + "cpu_baseline").

Workloads (BASELINE.json configs): "fwd" = configs[1] generator + sun-pose net (+ the Grad-CAM sweep
the reference runs inside its generator graph) forward, batch 32.
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

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
FWD_MFLOP_PER_IMG = 3220.3  # G + S + C (algorithmic 2*MAC of conv/dense contractions)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]: 32)")
    ap.add_argument("--workload", default="fwd", choices=["fwd"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    return ap.parse_args()


def dominant_kernel_roofline(torch, K, nets, batch, iters=200):
    """The res-block convolution (3x3, 128->128 on [B,8,32,128]: 12 of the generator's launches, 43% of its
    FLOPs) timed live with HIP events on the launch stream.  Algorithmic FLOPs per launch =
    2 * (B*8*32) * (3*3*128) * 128."""
    dev = nets.device
    x = torch.randn(batch, nets.h // 4, nets.w // 4, 128, device=dev)
    pw = nets.pk["gen.res.0.conv1"]
    bias = nets.gen["res.0.conv1.b"]
    y = torch.empty_like(x)
    for _ in range(10):
        K.conv2d(x, pw, bias, want_stats=True, compute=K.BF16, out=y)
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            K.conv2d(x, pw, bias, want_stats=True, compute=K.BF16, out=y)
    g.replay()
    torch.cuda.synchronize()
    e0.record(stream)
    g.replay()
    e1.record(stream)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flop = 2.0 * (batch * (nets.h // 4) * (nets.w // 4)) * (9 * 128) * 128
    achieved = flop / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": "conv_igemm_kernel (res-block 3x3 128->128, B=%d)" % batch,
            "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 5), "traffic": None,
            "avg_launch_us": round(us, 3), "flop_per_launch": flop}


def cpu_baseline(torch, gen, sun, batch_np):
    """CPU restatement (oracle/, NOT TensorFlow) of the same forward on this host's cores: bounded sample."""
    from oracle import step as ostep
    tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    g, s = tt(gen), tt(sun)
    ldr = torch.from_numpy(batch_np["ldr"])
    ostep.inference(g, s, ldr[:2])  # warm-up
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 12.0:
        ostep.inference(g, s, ldr)
        n += ldr.shape[0]
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d images (batches of %d) of the same synthetic workload through oracle/step.inference "
                      "(torch-CPU fp32 restatement, not TF2), %.1f s" % (n, ldr.shape[0], dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    params = importlib.import_module(PKG + ".params")
    synth = importlib.import_module(PKG + ".synth")
    engine = importlib.import_module(PKG + ".engine")
    K = importlib.import_module(PKG + ".kernels")

    gen = params.init_params(params.generator_spec(), 0)
    sun = params.init_params(params.sunpose_spec(), 1)
    batch_np = synth.make_batch(args.batch, seed=1234 + rank)
    nets = engine.Nets(gen, sun, device=dev, precise=False)
    ldr = torch.from_numpy(batch_np["ldr"]).to(dev)

    def step():
        return engine.generator_forward(nets, ldr, compute=K.BF16)

    # warm-up (eager) then capture one step into a hipGraph
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            out = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = step()
    run = graph.replay if graph is not None else step
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(out["y_final_lin"]).all()

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / dt
        res = {
            "metric": "images/sec (32x128 sky panoramas)", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_img": round(dt / imgs * world * 1e3, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic (seeded sky-dome + sun lobe, random-init weights)",
            "config": {"workload": "BASELINE configs[1]: generator + sunpose_net forward (incl. the Grad-CAM sweep "
                                   "and sun-radiance head of the generator graph), batch=%d per GPU, 32x128x3" % args.batch,
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": "replicas" if world > 1 else "single", "hipgraph": graph is not None},
            "algorithmic_tflops": round(value * FWD_MFLOP_PER_IMG * 1e6 / 1e12, 2),
        }
        res["roofline"] = dominant_kernel_roofline(torch, K, nets, args.batch)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(torch, gen, sun, batch_np)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
