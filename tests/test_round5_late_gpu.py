"""Round 5, late additions - claims under test:
(1) the plan variants only move WHEN a segment runs: the default step (target pass behind fwd_enc, the default segment merges) equals the
    unmerged plan (HDRSKY_PLAN_MERGE=), other merges,
    the round-4 plan (HDRSKY_VGG_TARGET_LATE=0) and a plan with extra dependencies (HDRSKY_PLAN_DEPS) bit for bit - gradients and
    updated weights; the loss values, sums by fp32 atomics, to rounding - over captured replays;
(2) the fused Dense update on a capped grid whose workgroups walk their k tiles (HDRSKY_FC_UPDATE_ROWS) writes the same w / ms / bf16
    images as the default launch, bit for bit; on two MFMA blocks per wave (HDRSKY_FC_UPDATE_NB=2, rounds 3-4) to the last bits of fp32;
(3) the resize-fused conv (upsample = 2: instantiations of their own since round 5, dispatch_tile_up) still matches the oracle when the
    table's tile for the layer has no such instantiation and the dispatcher falls back (a 7x7 conv to 3 channels at full resolution);
(4) HDRSKY_TILE_RULES injects a tile for matching layers only, and a conv's output does not depend on its tile (bit-identical y)."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _nets():
    params = pkg("params")
    return [params.init_params(params.generator_spec(), 0), params.init_params(params.sunpose_spec(), 1),
            params.init_params(params.discriminator_spec(), 2), params.init_params(params.vgg_spec(), 3)]


def _reload():
    pkg("hooks").reload()


def _run_plan(dev, monkeypatch, env):
    synth, trainer, K = pkg("synth"), pkg("trainer"), pkg("kernels")
    monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1")
    for k in ("HDRSKY_VGG_TARGET_LATE", "HDRSKY_PLAN_DEPS", "HDRSKY_PLAN_MERGE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _reload()
    tr = trainer.Trainer(*_nets(), device=dev, precise=False, compute=K.BF16, lr=2e-6)
    batches = [synth.make_batch(4, seed=500 + i) for i in range(2)]
    dv = lambda b: [torch.from_numpy(b[k]).to(dev) for k in ("ldr", "hdr_t", "sunpose_gt")]
    bufs = dv(batches[0])
    tr.capture(*bufs)
    order = [(n, deps) for n, _, deps, _ in tr._segs]
    out = []
    for b in batches:
        for dst, src in zip(bufs, dv(b)):
            dst.copy_(src)
        tr.replay()
        torch.cuda.synchronize()
        out.append((dict(tr.loss_dict()), tr.gs.grad.clone(), tr.ds.grad.clone(), tr.gs.flat.clone(), tr.ds.flat.clone()))
    return order, out


def test_plan_variants_are_bit_identical(dev, monkeypatch):
    base_order, base = _run_plan(dev, monkeypatch, {})
    names = [n for n, _ in base_order]
    assert "fwd_enc" in dict(base_order)["vgg_target"], "default plan: the target pass waits for fwd_enc"
    # default merges: the sun-side backward chain, the decoder / res-block pairs and the encoder pair are one segment each
    assert "bwd_sunpose" not in names and "wg_sunrad" not in names and "bwd_enc2" not in names and "bwd_res" in names and "wg_res" in names
    variants = (({"HDRSKY_VGG_TARGET_LATE": "0"}, lambda o: "fwd_enc" not in dict(o)["vgg_target"]),
                ({"HDRSKY_PLAN_MERGE": ""}, lambda o: {"bwd_sunpose", "wg_sunrad", "bwd_enc2"} <= {n for n, _ in o}),
                ({"HDRSKY_PLAN_MERGE": "bwd_dense+bwd_sunpose+bwd_sunrad+wg_sunrad,wg_dec+wg_res,bwd_dec+bwd_res"},
                 lambda o: "bwd_res" not in {n for n, _ in o} and "bwd_res" not in dict(o)["wg_dec"] and "bwd_dec" in dict(o)["wg_dec"]),
                ({"HDRSKY_PLAN_MERGE": "", "HDRSKY_PLAN_DEPS": "bwd_sunrad:disc_step,loss_vgg_b:loss_vgg"},
                 lambda o: "disc_step" in dict(o)["bwd_sunrad"] and "loss_vgg" in dict(o)["loss_vgg_b"]),
                ({"HDRSKY_PLAN_DEPS": "bwd_sunrad:disc_step"}, lambda o: "disc_step" in dict(o)["bwd_dense"]),     # (the merged chain inherits it)
                ({"HDRSKY_PLAN_MERGE": "grads_ready+apply,fwd_enc+zero"}, lambda o: "apply" not in {n for n, _ in o} and "bwd_sunpose" in {n for n, _ in o}))
    for env, check in variants:
        order, got = _run_plan(dev, monkeypatch, env)
        assert check(order), (env, order)
        for it, (a, b) in enumerate(zip(base, got)):
            for name in a[0]:      # (the loss VALUES are sums by fp32 atomics: equal to rounding, in any plan; the gradients are bit-reproducible)
                assert abs(a[0][name] - b[0][name]) <= 2e-6 * abs(a[0][name]) + 1e-9, (env, it, name, a[0][name], b[0][name])
            for k in range(1, 5):
                assert torch.equal(a[k], b[k]), (env, it, k)
    for k in ("HDRSKY_VGG_TARGET_LATE", "HDRSKY_PLAN_MERGE"):
        monkeypatch.delenv(k, raising=False)
    for bad in ({"HDRSKY_PLAN_DEPS": "disc_step:apply"}, {"HDRSKY_PLAN_MERGE": "bwd_dec+wg_dec"}, {"HDRSKY_PLAN_MERGE": "loss_adv+bwd_dec"}):
        with pytest.raises(ValueError):      # a dependency on a LATER segment / members on two streams / a segment of the stream in between
            _run_plan(dev, monkeypatch, bad)
    for k in ("HDRSKY_PLAN_DEPS", "HDRSKY_PLAN_MERGE"):
        monkeypatch.delenv(k, raising=False)
    _reload()


def test_fused_dense_update_grid_variants_are_bit_identical(dev, monkeypatch):
    K = pkg("kernels")
    g = torch.Generator(device=dev); g.manual_seed(11)
    Kd, N, M = 512, 1024, 32
    w0 = torch.randn(Kd, N, device=dev, generator=g) * 0.02
    ms0 = torch.rand(Kd, N, device=dev, generator=g) * 1e-4
    x = torch.randn(M, Kd, device=dev, generator=g)
    dy = torch.randn(M, N, device=dev, generator=g) * 1e-2

    def run(env):
        monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1")
        for k in ("HDRSKY_FC_UPDATE_ROWS", "HDRSKY_FC_UPDATE_NB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _reload()
        w, ms = w0.clone(), ms0.clone()
        pf = K.PackedFC(w, precise=False)
        db = torch.zeros(N, device=dev)
        K.rmsprop_fc_fused(w, ms, x, dy, pf, 1e-3, db=db)
        torch.cuda.synchronize()
        return w, ms, pf.pk_hi.clone(), pf.nat_hi.clone(), db

    ref = run({})
    assert not torch.equal(ref[0], w0)
    for env in ({"HDRSKY_FC_UPDATE_ROWS": "1"}, {"HDRSKY_FC_UPDATE_ROWS": "3"}, {"HDRSKY_FC_UPDATE_ROWS": "15"}):
        got = run(env)
        for k, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), (env, k)
    # two MFMA blocks per wave is ANOTHER instantiation: the same sums, but its update arithmetic is contracted into fused
    # multiply-adds differently - equal to the last bit or two of fp32, the bf16 images equal except where that bit decides a rounding
    ref2 = run({"HDRSKY_FC_UPDATE_NB": "2"})
    for k in (0, 1):
        err = float(((ref[k] - ref2[k]).abs() - 1e-6 * ref[k].abs()).max())      # (weights ~2e-2, slots ~1e-4: a few ulp of either)
        assert err < 2e-8, (k, err)
    assert float((ref[2].view(torch.int16) != ref2[2].view(torch.int16)).float().mean()) < 1e-3
    got = run({"HDRSKY_FC_UPDATE_NB": "2", "HDRSKY_FC_UPDATE_ROWS": "5"})
    for k, (a, b) in enumerate(zip(ref2, got)):
        assert torch.equal(a, b), ("NB=2, capped grid", k)
    for k in ("HDRSKY_FC_UPDATE_ROWS", "HDRSKY_FC_UPDATE_NB"):
        monkeypatch.delenv(k, raising=False)
    _reload()


def test_resize_fused_conv_falls_back_to_an_instantiated_tile(dev):
    """upsample = 2 with a layer class whose table entry (512 px x 16 ch for <= 16 output channels from 65536 pixels) has no
    resize-fused instantiation: the dispatcher takes the class's round-1 tile; result against torch (bilinear, half-pixel centres)."""
    K = pkg("kernels")
    g = torch.Generator(device=dev); g.manual_seed(5)
    B, H, W, C, F, k = 8, 16, 64, 32, 3, 7
    x = torch.randn(B, H, W, C, device=dev, generator=g)
    w = torch.randn(k, k, C, F, device=dev, generator=g) * 0.03
    bias = torch.randn(F, device=dev, generator=g)
    for precise, tol in ((False, 2e-2), (True, 2e-4)):
        pw = K.PackedConv(w, precise=precise)
        y, _ = K.conv2d(x, pw, bias, upsample=2, compute=K.BF16X3 if precise else K.BF16)
        xu = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=False)
        ref = torch.nn.functional.conv2d(xu, w.permute(3, 2, 0, 1).double(), bias.double(), padding=k // 2).permute(0, 2, 3, 1)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        assert y.shape == (B, 2 * H, 2 * W, F) and err < tol, (precise, err)


def test_tile_rules_select_by_layer_and_do_not_change_y(dev, monkeypatch):
    K, L = pkg("kernels"), pkg("_lib")
    g = torch.Generator(device=dev); g.manual_seed(9)
    x = (torch.randn(8, 32, 128, 64, device=dev, generator=g)).to(torch.bfloat16)
    pw = K.PackedConv(torch.randn(3, 3, 64, 64, device=dev, generator=g) * 0.05, precise=False)
    bias = torch.zeros(64, device=dev)
    monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1")
    monkeypatch.delenv("HDRSKY_TILE_RULES", raising=False); _reload()
    y0, _ = K.conv2d(x, pw, bias, compute=K.BF16, out_slope=0.0, out_bf16=True)
    d = K.conv_desc(8, 32, 128, 64, 64, 3, 3)
    d.compute = L.HDRSKY_BF16
    d.x_bf16 = d.y_bf16 = 1
    name0 = K.conv_kernel_name(d)
    # a rule that matches (64 -> 64, 3x3, M = 32768) and one that does not (another channel range)
    for rule, hit in (("64,64,32768,32768,64,64,3,-1=2,4,2,1,32,1", True), ("128,255,32768,32768,64,64,3,-1=2,4,2,1,32,1", False)):
        monkeypatch.setenv("HDRSKY_TILE_RULES", rule); _reload()
        y1, _ = K.conv2d(x, pw, bias, compute=K.BF16, out_slope=0.0, out_bf16=True)
        assert torch.equal(y0, y1), rule          # every output value is the same sum in the same order, whatever the tile
        name1 = K.conv_kernel_name(d)
        assert (name1 != name0) == hit and (("<2, 4, 2, 1, 32" in name1) == hit), (rule, name0, name1)
    monkeypatch.delenv("HDRSKY_TILE_RULES", raising=False); _reload()


def test_dense_finalisation_inside_the_launch_is_bit_identical(dev, monkeypatch):
    monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1"); monkeypatch.setenv("HDRSKY_FC_FIN", "1"); _reload()      # (off by default: not faster)
    """hdrsky_fc_fwd_fin / hdrsky_fc_dgrad_fin (the last workgroup of a column block adds the reduction slices) against
    hdrsky_fc_fwd / _dgrad + hdrsky_fc_finalize: bit for bit, over many back-to-back launches on changing inputs (a stale read of
    another workgroup's slice would show as a mismatch), eager and from a replayed graph, both compute modes, ragged row counts."""
    K = pkg("kernels")
    g = torch.Generator(device=dev); g.manual_seed(21)
    Kd, N = 2048, 1024
    w = torch.randn(Kd, N, device=dev, generator=g) * 0.03
    bias = torch.randn(N, device=dev, generator=g)
    for precise, compute in ((False, K.BF16), (True, K.BF16X3)):
        pf = K.PackedFC(w, precise=precise)
        for M in (1, 7, 32):
            for it in range(40):
                x = torch.randn(M, Kd, device=dev, generator=g)
                dy = torch.randn(M, N, device=dev, generator=g)
                mask = torch.randn(M, Kd, device=dev, generator=g)
                zw = torch.ones(1, dtype=torch.int32, device=dev)
                a = K.fc_fwd_fin(x, pf, compute, bias, relu=True, zero_word=zw)
                b = K.fc_finalize(K.fc_fwd(x, pf, compute), bias, relu=True)
                assert torch.equal(a, b) and int(zw.item()) == 0, (precise, M, it)
                a = K.fc_dgrad_fin(dy, pf, compute, mask_src=mask)
                b = K.fc_finalize(K.fc_dgrad(dy, pf, compute), None, relu=False, mask_src=mask)
                assert torch.equal(a, b), (precise, M, it, "dgrad")
    # replayed graph: the tickets reset themselves
    pf = K.PackedFC(w, precise=False)
    x = torch.randn(32, Kd, device=dev, generator=g)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y0 = K.fc_fwd_fin(x, pf, K.BF16, bias, relu=True)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            y = K.fc_fwd_fin(x, pf, K.BF16, bias, relu=True)
            d = K.fc_dgrad_fin(y, pf, K.BF16)
        for it in range(30):
            x.copy_(torch.randn(32, Kd, device=dev, generator=g))
            gr.replay()
            s.synchronize()
            ref = K.fc_finalize(K.fc_fwd(x, pf, K.BF16), bias, relu=True)
            assert torch.equal(y, ref), it
            assert torch.equal(d, K.fc_finalize(K.fc_dgrad(ref, pf, K.BF16))), it
    # HDRSKY_FC_FIN=0: the two launches
    monkeypatch.setenv("HDRSKY_FC_FIN", "0"); _reload()
    assert not pkg("hooks").H.fc_fin
    assert torch.equal(K.fc_fwd_fin(x, pf, K.BF16, bias, relu=True), K.fc_finalize(K.fc_fwd(x, pf, K.BF16), bias, relu=True))
    monkeypatch.delenv("HDRSKY_FC_FIN", raising=False); _reload()
