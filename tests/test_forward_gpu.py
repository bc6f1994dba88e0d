"""GPU parity of the whole generator graph (inference.py:81-115 / train.py:239-299 test mode) vs the oracle."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import step as ostep
from util import assert_close, rel_max, rel_rms, to_np

pytestmark = pytest.mark.gpu


def _setup(dev, B):
    params, synth, engine = pkg("params"), pkg("synth"), pkg("engine")
    gen = params.init_params(params.generator_spec(), 0)
    sun = params.init_params(params.sunpose_spec(), 1)
    batch = synth.make_batch(B, seed=1234)
    nets = engine.Nets(gen, sun, device=dev)
    tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    return engine, nets, tt(gen), tt(sun), batch


def psnr(a, b):
    a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
    peak = np.abs(b).max()
    return 10 * np.log10(peak * peak / ((a - b) ** 2).mean())


@pytest.mark.parametrize("B", [1, 3])
def test_generator_graph_inference_mode(dev, B):
    K = pkg("kernels")
    engine, nets, gen, sun, batch = _setup(dev, B)
    ref = ostep.inference(gen, sun, torch.from_numpy(batch["ldr"]))
    ldr = torch.from_numpy(batch["ldr"]).to(dev)
    out = engine.generator_forward(nets, ldr, compute=K.BF16X3)
    # Stage-wise checks (fp32-class BF16X3 contractions; tolerance relative to each tensor's max)
    for k in ("res_out", "sunpose_cmf", "sun_cam1", "sun_cam2", "sun_cam3", "gamma", "beta", "sun_rad_lin", "alpha_c3",
              "y_final_gamma", "y_final_lin"):
        print("%-14s rel max %.3e  rel rms %.3e" % (k, rel_max(out[k], ref[k]), rel_rms(out[k], ref[k])))
    assert_close(out["res_out"], ref["res_out"], 1e-3, "res_out")
    assert_close(out["sunpose_cmf"], ref["sunpose_cmf"], 2e-3, "cmf")
    # Grad-CAM maps: the backward sweep routes gradients through max-pool arg-maxes.  Where neighbouring
    # activations tie to within rounding (flat / clipped image regions give bitwise-equal conv outputs),
    # the arg-max - hence which pixel receives the gradient - is decided by last-bit noise, in the oracle
    # as much as here (measured: feeding the ORACLE's own tensors through hdrsky_norm_act_bwd reproduces
    # autograd to 7e-7 except in the one or two channels per sample that contain such a tie).  The maps
    # therefore get a looser tolerance than the smooth tensors.  The claim itself is pinned where it can be separated:
    # tests/test_ops_gpu.py::test_norm_act_bwd_pool_routing_tight_off_the_tie_channels asserts 1e-4 on every
    # (sample, channel) slice without a tie and the loose bound only on the slices the oracle flags.
    for k in ("sun_cam1", "sun_cam2", "sun_cam3"):
        assert_close(out[k], ref[k], 5e-2, k)
        assert rel_rms(out[k], ref[k]) < 3e-2, k
    assert_close(out["gamma"], ref["gamma"], 1e-4, "gamma"); assert_close(out["beta"], ref["beta"], 1e-4, "beta")
    assert_close(out["sun_rad_lin"], ref["sun_rad_lin"], 2e-3, "sun_rad_lin")
    assert_close(out["alpha_c3"], ref["alpha_c3"], 5e-3, "alpha")   # slope 1/0.12 on top of the exp() of the decompression
    assert_close(out["y_final_gamma"], ref["y_final_gamma"], 1e-3, "y_final_gamma")
    assert_close(out["y_final_lin"], ref["y_final_lin"], 5e-3, "y_final_lin")
    # fast path: single bf16 product.  PSNR of the gamma-domain output vs the fp32 oracle
    out16 = engine.generator_forward(nets, ldr, compute=K.BF16)
    p = psnr(out16["y_final_gamma"], ref["y_final_gamma"])
    print("bf16 y_final_gamma PSNR vs oracle: %.1f dB, rel max %.3e" % (p, rel_max(out16["y_final_gamma"], ref["y_final_gamma"])))
    assert p > 40.0


def test_generator_graph_picks_gt_bin(dev):
    """train.py:265-267: y_c = cmf[b, argmax(sunpose_gt[b])]."""
    K = pkg("kernels")
    engine, nets, gen, sun, batch = _setup(dev, 2)
    gt = torch.from_numpy(batch["sunpose_gt"])
    req = {k: v.clone().requires_grad_(True) for k, v in sun.items()}
    ref = ostep.generator_graph(gen, req, torch.from_numpy(batch["ldr"]), y_index=gt.argmax(dim=1), training=False)
    out = engine.generator_forward(nets, torch.from_numpy(batch["ldr"]).to(dev), pick_src=gt.to(dev), compute=K.BF16X3)
    for k in ("sun_cam1", "sun_cam2", "sun_cam3"):
        assert_close(out[k], ref[k].detach(), 1e-1, k)
        # (picking a low-probability bin makes the maps even smaller and more tie-dominated than the max-bin case)
        assert rel_rms(out[k], ref[k].detach()) < 6e-2, k
    assert_close(out["y_final_gamma"], ref["y_final_gamma"].detach(), 1e-3, "y_final_gamma")


@pytest.mark.parametrize("da", [False, "all"])
def test_captured_forward_replays_match_eager(dev, da):
    """The forward pass as ONE hipGraph (what bench.py times: two branches forked onto two streams inside the capture)
    replayed several times, with eager allocations in between, against the eager pass: every output bit for bit."""
    K = pkg("kernels")
    engine, nets, gen, sun, batch = _setup(dev, 8)
    ldr = torch.from_numpy(batch["ldr"]).to(dev)
    fn = lambda: engine.generator_forward(nets, ldr, compute=K.BF16, distortion_aware=da)
    ref = {k: v.clone() for k, v in fn().items() if torch.is_tensor(v)}
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with pkg("kernels").no_gc(), torch.cuda.graph(g):
        out = fn()
    for it in range(4):
        g.replay()
        torch.cuda.synchronize()
        spare = [torch.empty(1 << 20, device=dev).normal_() for _ in range(4)]
        for k, v in ref.items():
            assert torch.equal(out[k], v), (it, k)
        del spare


@pytest.mark.parametrize("da", [False, "all"])
def test_forward_graphs_on_two_streams_match_eager(dev, da):
    """engine.ForwardGraphs (what bench.py's `fwd` times since round 5: one hipGraph per branch on its own stream, the tail
    behind an event) replayed several times, with the input rewritten in between, against the eager pass: bit for bit."""
    K = pkg("kernels")
    engine, nets, gen, sun, batch = _setup(dev, 8)
    ldr = torch.from_numpy(batch["ldr"]).to(dev)
    other = torch.from_numpy(pkg("synth").make_batch(8, seed=77)["ldr"]).to(dev)
    fn = lambda x: {k: v.clone() for k, v in engine.generator_forward(nets, x, compute=K.BF16, distortion_aware=da).items() if torch.is_tensor(v)}
    ref_a, ref_b = fn(ldr.clone()), fn(other)
    torch.cuda.synchronize()
    x = ldr.clone()
    fg = engine.ForwardGraphs(nets, x, compute=K.BF16, distortion_aware=da)
    for it in range(4):
        x.copy_(ldr if it % 2 == 0 else other)
        out = fg.replay()
        torch.cuda.synchronize()
        for k, v in (ref_a if it % 2 == 0 else ref_b).items():
            assert torch.equal(out[k], v), (it, k)
