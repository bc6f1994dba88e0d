"""CPU tests that pin the oracle to things OTHER than itself ("What's missing" item 1).

(a) The reference's only held data - the two Radiance images `figure/1_gt.hdr` and `DataGeneration/test.hdr`, committed
    as data fixtures `tests/golden/ref_1_gt.hdr` / `ref_test.hdr` (cv2-written new-style RLE files) - decode through
    `hdr_io.read_hdr` to the statistics SURVEY.md section 8c recorded with an independent throw-away reader, and
    `write_hdr -> read_hdr` is idempotent on them.
(b) `oracle/tfsem.py`'s conv2d / gaussian_filter2d / resize_bilinear agree with independent third-party code present in
    the image (scipy.signal, scipy.ndimage, Pillow), the way `oracle/jpeg.py` is pinned to libjpeg.  What this pins is
    the arithmetic (correlation not convolution, HWIO filter layout, reflect-without-edge-repeat padding, half-pixel
    sample positions); the TF-side choice of those conventions stays as stated in SURVEY.md section 8c appendix.
"""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tfsem as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_HDR = {
    # file: (max, mean, median, sun row, sun col)   -- SURVEY.md section 8c "Fixtures available in the reference"
    "ref_1_gt.hdr": (2400.0, 0.497, 0.036, 25, 63),
    "ref_test.hdr": (644.0, 3.77, None, 16, 62),
}


@pytest.mark.parametrize("name", sorted(REF_HDR))
def test_reference_hdr_fixture_decodes_to_survey_statistics(name, tmp_path):
    io = pkg("hdr_io")
    mx, mean, median, srow, scol = REF_HDR[name]
    path = os.path.join(GOLD, name)
    raw = open(path, "rb").read()
    assert raw.startswith(b"#?RADIANCE") and b"-Y 32 +X 128" in raw
    body = raw[raw.index(b"+X 128\n") + 7:]
    assert body[:4] == b"\x02\x02\x00\x80", "fixture must exercise the new-style RLE path"
    img = io.read_hdr(path)
    assert img.shape == (32, 128, 3) and img.dtype == np.float32 and np.isfinite(img).all() and img.min() >= 0
    assert float(img.max()) == mx
    assert abs(float(img.mean()) - mean) <= 5e-3 * mean + 5e-4
    if median is not None:
        assert abs(float(np.median(img)) - median) <= 1e-3
    r, c = np.unravel_index(img.max(-1).argmax(), img.shape[:2])
    assert (int(r), int(c)) == (srow, scol)
    # write -> read is idempotent on already RGBE-quantised data (flat writer, RLE/flat reader)
    p2 = os.path.join(str(tmp_path), "again.hdr")
    io.write_hdr(p2, img)
    back = io.read_hdr(p2)
    assert np.array_equal(back, img)
    # and the RGBE bytes themselves survive the round trip
    assert np.array_equal(io.float_to_rgbe(back[..., ::-1]), io.float_to_rgbe(img[..., ::-1]))


def _tf_same_pad(n, k, s):
    """TF documentation formula, written independently of oracle/tfsem.same_pad."""
    out = (n + s - 1) // s
    need = (out - 1) * s + k - n
    need = need if need > 0 else 0
    return need // 2, need - need // 2


@pytest.mark.parametrize("k,stride,h,w", [(3, 1, 8, 12), (3, 2, 8, 12), (4, 2, 8, 12), (4, 1, 6, 9), (7, 1, 9, 11), (3, 2, 7, 9)])
def test_conv2d_against_scipy_signal(k, stride, h, w):
    """tf.nn.conv2d is a CORRELATION with an HWIO filter; SAME pads (total//2, rest).  scipy.signal.correlate2d does
    the arithmetic (independent code: no torch, no im2col), the pad amounts come from the TF formula above."""
    from scipy.signal import correlate2d
    rng = np.random.default_rng(k * 10 + stride)
    cin, cout = 3, 4
    x = rng.standard_normal((2, h, w, cin)).astype(np.float32)
    wgt = rng.standard_normal((k, k, cin, cout)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    got = T.conv2d(torch.from_numpy(x), torch.from_numpy(wgt), torch.from_numpy(b), stride, "SAME").numpy()
    (pt, pb), (pl, pr) = _tf_same_pad(h, k, stride), _tf_same_pad(w, k, stride)
    xp = np.pad(x.astype(np.float64), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    ref = np.zeros(got.shape, np.float64)
    for n in range(2):
        for co in range(cout):
            acc = 0.0
            for ci in range(cin):
                acc = acc + correlate2d(xp[n, :, :, ci], wgt[:, :, ci, co].astype(np.float64), mode="valid")
            ref[n, :, :, co] = acc[::stride, ::stride] + b[co]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    if k % 2 == 1 and stride == 1:
        # odd kernel, stride 1: scipy's own mode='same' centring must give the same answer (padding cross-check)
        same = sum(correlate2d(x[0, :, :, ci].astype(np.float64), wgt[:, :, ci, 0].astype(np.float64), mode="same")
                   for ci in range(cin)) + b[0]
        assert np.abs(got[0, :, :, 0] - same).max() <= 2e-5 * np.abs(same).max()


@pytest.mark.parametrize("sigma", [T.DOG_SIGMA_BASE, T.DOG_SIGMAS_1[0], T.DOG_SIGMAS_2[3]])
def test_gaussian_filter_against_scipy_ndimage(sigma):
    """tfa.image.gaussian_filter2d(filter_shape=3, sigma, 'REFLECT') = separable 3-tap correlation with
    tf.pad(REFLECT) borders.  tf.pad REFLECT does not repeat the edge sample = scipy.ndimage mode='mirror'."""
    from scipy import ndimage
    rng = np.random.default_rng(7)
    x = rng.random((2, 9, 13, 3)).astype(np.float32)
    got = T.gaussian_filter2d_3x3(torch.from_numpy(x), sigma).numpy()
    k = np.exp(-np.array([1.0, 0.0, 1.0]) / (2.0 * sigma * sigma))
    k = k / k.sum()                                        # = softmax(-x^2 / (2 sigma^2)) over x in {-1, 0, 1}
    ref = ndimage.correlate1d(x.astype(np.float64), k, axis=1, mode="mirror")
    ref = ndimage.correlate1d(ref, k, axis=2, mode="mirror")
    assert np.abs(got - ref).max() <= 2e-6
    # the edge-repeating alternative ('reflect' in scipy = SYMMETRIC in tf.pad) is measurably different: the test can tell
    alt = ndimage.correlate1d(ndimage.correlate1d(x.astype(np.float64), k, axis=1, mode="reflect"), k, axis=2, mode="reflect")
    assert np.abs(got - alt).max() > 1e-3


@pytest.mark.parametrize("h,w,oh,ow", [(8, 32, 16, 64), (4, 16, 8, 32), (5, 7, 10, 14), (8, 32, 32, 128), (6, 9, 13, 20)])
def test_resize_bilinear_against_pillow_and_scipy(h, w, oh, ow):
    """tf.image.resize(BILINEAR) in TF2 = half-pixel centres, no antialias.  For up-scaling Pillow's BILINEAR (triangle
    filter of support 1 centred at (dst+0.5)*in/out, clipped and renormalised at the borders) and
    scipy.ndimage.zoom(order=1, grid_mode=True, mode='nearest') are the same map."""
    from PIL import Image
    from scipy import ndimage
    rng = np.random.default_rng(h * 100 + w)
    x = rng.random((1, h, w, 2)).astype(np.float32)
    got = T.resize_bilinear(torch.from_numpy(x), oh, ow).numpy()
    for c in range(2):
        pil = np.asarray(Image.fromarray(x[0, :, :, c], mode="F").resize((ow, oh), Image.BILINEAR))
        assert np.abs(got[0, :, :, c] - pil).max() <= 2e-6
        z = ndimage.zoom(x[0, :, :, c].astype(np.float64), (oh / h, ow / w), order=1, grid_mode=True, mode="nearest")
        assert z.shape == (oh, ow) and np.abs(got[0, :, :, c] - z).max() <= 2e-6
    # an align-corners map would differ: the test is sensitive to the convention
    ac = torch.nn.functional.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), size=(oh, ow), mode="bilinear",
                                         align_corners=True).permute(0, 2, 3, 1).numpy()
    assert np.abs(got - ac).max() > 1e-3


def test_maxpool_and_dense_against_numpy():
    """2x2/2 max pool on even dims (no padding) and Keras Dense x@W+b, against plain numpy reshapes."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 6, 8, 3)).astype(np.float32)
    got = T.maxpool2x2(torch.from_numpy(x)).numpy()
    ref = x.reshape(2, 3, 2, 4, 2, 3).max(axis=(2, 4))
    assert np.array_equal(got, ref)
    f = rng.standard_normal((2, 10)).astype(np.float32)
    k = rng.standard_normal((10, 4)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    assert np.allclose(T.dense(torch.from_numpy(f), torch.from_numpy(k), torch.from_numpy(b)).numpy(), f @ k + b, atol=1e-5)
    # Flatten on NHWC is (h, w, c) row-major
    assert np.array_equal(T.flatten_nhwc(torch.from_numpy(x)).numpy(), x.reshape(2, -1))


def test_reference_hdr_inputs_run_through_the_oracle():
    """The two reference images as INPUTS of the restated inference graph (B=2): finite output, the sun stays the
    brightest region of the reconstruction input, and the committed end-to-end golden (generated by this repo's oracle,
    labelled so in tests/golden/make_golden.py) still matches."""
    io, params, synth = pkg("hdr_io"), pkg("params"), pkg("synth")
    from oracle import step
    hdr = np.stack([io.read_hdr(os.path.join(GOLD, n)) for n in sorted(REF_HDR)])
    batch = synth.batch_from_hdr(hdr)
    assert batch["ldr"].shape == (2, 32, 128, 3) and batch["ldr"].min() >= 0 and batch["ldr"].max() <= 1
    assert np.allclose(batch["sunpose_gt"].sum(1), 1.0, atol=1e-5)
    gen = params.init_params(params.generator_spec(), 0)
    sun = params.init_params(params.sunpose_spec(), 1)
    tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    torch.set_num_threads(4)
    out = step.inference(tt(gen), tt(sun), torch.from_numpy(batch["ldr"]))
    y = out["y_final_lin"].numpy()
    assert y.shape == (2, 32, 128, 3) and np.isfinite(y).all() and y.min() >= 0
    g = np.load(os.path.join(GOLD, "ref_hdr_forward.npz"))
    for k in ("y_final_gamma", "sunpose_cmf", "gamma", "beta"):
        ref = g[k]
        assert np.abs(out[k].numpy() - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-30), k


def test_norms_optimizers_and_kl_against_torch_library_code():
    """(c) The oracle's hand-written InstanceNorm / BatchNorm / RMSprop / Adam / KL formulas against the library
    implementations that ship with torch (torch.nn.functional, torch.optim - independent code of the same published
    algorithms).  What stays a TF-side assumption: eps = 1e-3 and its place inside the root (Keras / tfa defaults), the
    Bessel-corrected moving variance of the fused BatchNorm path, eps OUTSIDE the root in both optimizers."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 6, 10, 8, generator=g) * 2.0 + 0.5
    gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g)
    nchw = x.permute(0, 3, 1, 2)
    ref = F.instance_norm(nchw, weight=gamma, bias=beta, eps=1e-3).permute(0, 2, 3, 1)
    assert torch.allclose(T.instance_norm(x, gamma, beta), ref, rtol=1e-5, atol=1e-5)
    # BatchNorm, training: batch statistics + moving averages (torch's momentum is 1 - Keras' momentum, and torch feeds
    # the running variance with the unbiased estimate - the same convention the oracle assumes for the fused TF path)
    mm, mv = torch.randn(8, generator=g), torch.rand(8, generator=g) + 0.5
    y, nm, nv = T.batch_norm(x, gamma, beta, mm, mv, training=True)
    rm, rv = mm.clone(), mv.clone()
    ref = F.batch_norm(nchw, rm, rv, weight=gamma, bias=beta, training=True, momentum=0.01, eps=1e-3).permute(0, 2, 3, 1)
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(nm, rm, rtol=1e-6, atol=1e-6) and torch.allclose(nv, rv, rtol=1e-6, atol=1e-6)
    ye, _, _ = T.batch_norm(x, gamma, beta, mm, mv, training=False)
    ref = F.batch_norm(nchw, mm, mv, weight=gamma, bias=beta, training=False, eps=1e-3).permute(0, 2, 3, 1)
    assert torch.allclose(ye, ref, rtol=1e-5, atol=1e-5)
    # RMSprop (rho 0.9, eps 1e-7 outside the root, no momentum, not centred) and Adam over four steps
    w0 = torch.randn(50, generator=g)
    grads = [torch.randn(50, generator=g) * 10.0 ** float(e) for e in (-3, 0, -1, 1)]
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.RMSprop([p], lr=1e-3, alpha=0.9, eps=1e-7)
    w, ms = w0.clone(), torch.zeros(50)
    for gr in grads:
        p.grad = gr.clone(); opt.step()
        w, ms = T.rmsprop_update(w, gr, ms, 1e-3)
        assert torch.allclose(w, p.detach(), rtol=1e-6, atol=1e-7)
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-7)
    w, m, v = w0.clone(), torch.zeros(50), torch.zeros(50)
    for step, gr in enumerate(grads, 1):
        gr = gr.sign() * gr.abs().clamp_min(0.05)
        p.grad = gr.clone(); opt.step()
        w, m, v = T.adam_update(w, gr, m, v, 1e-3, step)
        # Keras folds the bias correction into lr_t, i.e. its eps is torch's eps scaled by sqrt(1 - b2^t): the two forms
        # differ by O(eps / sqrt(v)) - compared here on gradients large enough for that to vanish
        assert torch.allclose(w, p.detach(), rtol=1e-5, atol=1e-6), step
    # KLDivergence (Keras clips both arguments to [1e-7, 1]; sum over the last axis, mean over the batch)
    yt = torch.softmax(torch.randn(4, 64, generator=g) * 3, -1); yp = torch.softmax(torch.randn(4, 64, generator=g), -1)
    ref = F.kl_div(yp.clamp(1e-7, 1).log(), yt.clamp(1e-7, 1), reduction="none").sum(-1).mean()
    assert abs(float(T.kl_divergence(yt, yp)) - float(ref)) <= 1e-6 * abs(float(ref))
