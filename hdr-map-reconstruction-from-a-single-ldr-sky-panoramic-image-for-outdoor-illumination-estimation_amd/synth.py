"""Seeded synthetic inputs shaped like the reference's training batches (host-side numpy).

There is no dataset, no ``dorfCurves.txt`` and no ``vgg16.npy`` in this environment, so the
bench / tests use an analytic sky-dome + sun lobe that mimics what train.py feeds the step:
  hdr_t            [B,H,W,3] fp32 BGR, mean normalised to 0.5 then exposure-scaled, plus the
                   signal-dependent + constant Gaussian noise of the augmentation
                   (train.py:109-110 ``0.5*hdr/(mean+1e-6)``, utils.py:86-91 ``2**U(-3,3)``, train.py:66-74)
  jpeg_img_float   [B,H,W,3] fp32 on the k/255 lattice (train.py:79-92: clip, CRF, 8-bit quantise;
                   the CRF LUT + JPEG round trip are replaced by a 1/2.2 gamma)
  sunpose_gt       [B,H*W] von-Mises-Fisher pmf over the sky bins (train.py:42-52 with the bin
                   directions of tf_utils.sunpose_init, tf_utils.py:112-129, and
                   tf_utils.sphere2world, tf_utils.py:95-110)
"""
import numpy as np

KAPPA = 80.0  # train.py:42


def sunpose_bins(h, w):
    """Unit vectors of the h*w sky bins: tf_utils.sunpose_init (tf_utils.py:112-129) for i in range(h*w)."""
    i = np.arange(h * w, dtype=np.float64)
    row = np.floor(i / w)
    x = ((i + 1.0) - row * w - 1.0) * (360.0 / w) + (360.0 / (w * 2.0))
    y = row * (90.0 / h) + (90.0 / (2.0 * h))
    phi = y * (np.pi / 180.0)
    theta = (x - 180.0) * (np.pi / 180.0)
    return np.stack([np.cos(phi) * np.cos(theta), np.sin(phi), np.cos(phi) * np.sin(theta)], axis=1)


def sphere2world(x, y, h, w):
    """tf_utils.sphere2world (tf_utils.py:95-110), skydome=True."""
    unit_w = 2.0 * np.pi / w
    unit_h = np.pi / (h * 2)
    theta = (x - 0.5 * w) * unit_w
    phi = (h - y) * unit_h
    return np.array([np.cos(phi) * np.cos(theta), np.sin(phi), np.cos(phi) * np.sin(theta)])


def vmf_target(azimuth, elevation, h, w, kappa=KAPPA):
    """train.py:42-52 ``vMF``: exp(kappa * <bin, sun>) normalised to sum 1."""
    v = sphere2world(azimuth, elevation, h, w)
    pdf = np.exp(kappa * (sunpose_bins(h, w) @ v))
    return (pdf / pdf.sum()).astype(np.float32)


def make_batch(batch, h=32, w=128, seed=1234):
    """Returns dict(hdr_t, ldr, sunpose_gt) of np.float32 arrays (BGR channel order)."""
    rng = np.random.default_rng(seed)
    ys = np.arange(h, dtype=np.float64).reshape(h, 1)
    bins = sunpose_bins(h, w).reshape(h, w, 3)
    azimuth = w * 0.5 - 1.0  # AZIMUTH_gt, train.py:32
    hdr = np.empty((batch, h, w, 3), np.float64)
    gt = np.empty((batch, h * w), np.float32)
    for b in range(batch):
        tint = rng.uniform(0.8, 1.2, size=3)
        sky = (0.2 + 0.6 * (ys / h)) * np.ones((1, w))
        elev_row = float(rng.integers(0, h))
        amp = 10.0 ** rng.uniform(1.0, 3.4)
        sun_dir = sphere2world(azimuth, elev_row, h, w)
        lobe = amp * np.exp(KAPPA * (bins @ sun_dir - 1.0))
        img = sky[:, :, None] * tint[None, None, :] + lobe[:, :, None]
        img = 0.5 * img / (img.mean() + 1e-6)
        img = img * 2.0 ** rng.uniform(-3.0, 3.0)
        # Poisson-like + Gaussian sensor noise of the reference's augmentation (train.py:66-74)
        sigma_s = 0.08 / 6.0 * rng.uniform(0.0, 1.0, size=(1, 1, 3))
        sigma_c = 0.005 * rng.uniform(0.0, 1.0, size=(1, 1, 3))
        img = img + rng.standard_normal(img.shape) * sigma_s * img + sigma_c * rng.standard_normal(img.shape)
        img = np.maximum(img, 0.0)
        hdr[b] = img
        gt[b] = vmf_target(azimuth, elev_row, h, w)
    hdr = hdr.astype(np.float32)
    ldr = np.round(255.0 * np.clip(hdr, 0.0, 1.0) ** (1.0 / 2.2)) / 255.0
    return dict(hdr_t=hdr, ldr=ldr.astype(np.float32), sunpose_gt=gt)


def batch_from_hdr(hdr, exposure=1.0):
    """A training / inference batch from real HDR panoramas [B,H,W,3] (BGR, linear radiance), e.g. the reference's two
    Radiance images: the mean normalisation of train.py:109-110 (``0.5*hdr/(mean+1e-6)``), a fixed exposure instead of
    the random one (utils.py:86-91), the same gamma + 8-bit stand-in for CRF/JPEG as make_batch, and the vMF target
    (train.py:42-52) centred on the brightest pixel.  Returns dict(hdr_t, ldr, sunpose_gt) of np.float32 arrays."""
    hdr = np.asarray(hdr, np.float64)
    b, h, w, _ = hdr.shape
    out = np.empty_like(hdr)
    gt = np.empty((b, h * w), np.float32)
    for i in range(b):
        img = 0.5 * hdr[i] / (hdr[i].mean() + 1e-6) * float(exposure)
        out[i] = img
        row, col = np.unravel_index(img.max(-1).argmax(), (h, w))
        gt[i] = vmf_target(float(col), float(row), h, w)
    out = out.astype(np.float32)
    ldr = np.round(255.0 * np.clip(out, 0.0, 1.0) ** (1.0 / 2.2)) / 255.0
    return dict(hdr_t=out, ldr=ldr.astype(np.float32), sunpose_gt=gt)


def gamma_crf(n_curves=1, k=1024, gamma=2.2):
    """Stand-in for the DoRF response curves (dorfCurves.txt is not available): k samples of x^(1/gamma) on [0,1]."""
    x = np.linspace(0.0, 1.0, k)
    return np.tile((x ** (1.0 / gamma)).astype(np.float32)[None], (n_curves, 1))


def load_dorf(path, n_train=175):
    """utils.getDoRF (utils.py:105-116): `dorfCurves.txt` holds six lines per camera response curve - name, type, a label
    and the irradiance samples, a label and the BRIGHTNESS samples; the sixth line of every record (index 5) is the
    response curve as k whitespace-separated floats.  The first 175 curves train, the rest (26 in the published file)
    test.  Returns (train [n,k] float32, test [m,k] float32)."""
    with open(path, "r") as f:
        lines = [ln.strip() for ln in f.readlines()]
    rows = [lines[i + 5].split() for i in range(0, len(lines) - 5, 6)]
    if not rows or len({len(r) for r in rows}) != 1:
        raise ValueError("%s: not a DoRF curve file (six lines per curve, the sixth holding its samples)" % path)
    crf = np.asarray(rows, dtype=np.float32)
    return crf[:n_train], crf[n_train:]


def pick_crf(curves, batch, seed):
    """train.py:58: one response curve per sample, drawn uniformly from the list."""
    idx = np.random.default_rng(int(seed) ^ 0x5EED).integers(0, len(curves), size=batch)
    return np.ascontiguousarray(curves[idx])


def make_batch_device(batch, h=32, w=128, seed=1234, device="cuda", crf=None, jpeg=True):
    """The same synthetic distribution as make_batch, produced on the GPU: analytic sky-dome + sun lobe (torch ops on
    the device - test/bench plumbing), then the reference's augmentation and target construction in libhdrsky
    (kernels.ldr_synth = train.py:54-85, kernels.jpeg_roundtrip = train.py:86-92 when `jpeg`,
    kernels.vmf_target = train.py:42-52).  Returns dict of CUDA tensors (hdr_t, ldr, sunpose_gt).  Different random
    stream than make_batch (torch generator instead of numpy; make_batch has no JPEG step)."""
    import torch
    from . import kernels as K
    dev = torch.device(device)
    g = torch.Generator(device=dev); g.manual_seed(int(seed))
    u = lambda *shape: torch.rand(*shape, device=dev, generator=g)
    ys = torch.arange(h, device=dev, dtype=torch.float32).view(1, h, 1, 1)
    tint = 0.8 + 0.4 * u(batch, 1, 1, 3)
    elev = torch.floor(u(batch) * h)
    amp = 10.0 ** (1.0 + 2.4 * u(batch))
    bins = torch.from_numpy(sunpose_bins(h, w).astype(np.float32)).to(dev).view(1, h, w, 3)
    azimuth = w * 0.5 - 1.0
    theta = (azimuth - 0.5 * w) * (2.0 * np.pi / w)
    phi = (h - elev) * (np.pi / (h * 2))
    sun = torch.stack([torch.cos(phi) * np.cos(theta), torch.sin(phi), torch.cos(phi) * np.sin(theta)], dim=1).view(batch, 1, 1, 3)
    lobe = amp.view(batch, 1, 1, 1) * torch.exp(KAPPA * ((bins * sun).sum(-1, keepdim=True) - 1.0))
    img = (0.2 + 0.6 * ys / h) * tint + lobe
    img = (0.5 * img / (img.mean(dim=(1, 2, 3), keepdim=True) + 1e-6)).contiguous()          # train.py:109-110
    t = 2.0 ** (-3.0 + 6.0 * u(batch))                                                          # utils.py:86-91
    sigma_s, sigma_c = 0.08 / 6.0 * u(batch, 3), 0.005 * u(batch, 3)                            # train.py:66-68
    n_s = torch.randn(batch, h, w, 3, device=dev, generator=g)
    n_c = torch.randn(batch, h, w, 3, device=dev, generator=g)
    if crf is None:
        crf = torch.from_numpy(gamma_crf(batch)).to(dev)
    elif isinstance(crf, np.ndarray):      # a curve list (load_dorf): one curve per sample (train.py:58)
        crf = torch.from_numpy(pick_crf(crf, batch, seed)).to(dev)
    hdr_t, ldr = K.ldr_synth(img, t, sigma_s.contiguous(), sigma_c.contiguous(), n_s, n_c, crf)
    if jpeg:
        K.jpeg_roundtrip(ldr, order="bgr", out=ldr)
    return dict(hdr_t=hdr_t, ldr=ldr, sunpose_gt=K.vmf_target(elev.contiguous(), azimuth, h, w, KAPPA))
