"""Shared test helpers: error metrics and tolerances.

Tolerances (floating point path; stated here once, used by every GPU parity test):
  TOL_X3   : HDRSKY_BF16X3 contractions (three bf16 MFMA products of hi/lo split fp32 operands,
             fp32 accumulate) vs the fp32 CPU oracle: max|err| <= 2e-4 * max|ref| per operator.
  TOL_BF16 : HDRSKY_BF16 (single bf16 product) vs the oracle: max|err| <= 2.5e-2 * max|ref|
             and rms(err) <= 6e-3 * rms(ref) per operator.
  TOL_F32  : pure fp32 pointwise / reduction kernels: 2e-5 * max|ref|.
"""
import numpy as np
import torch

TOL_X3 = 2e-4
TOL_BF16_MAX = 2.5e-2
TOL_BF16_RMS = 6e-3
TOL_F32 = 2e-5


def to_np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def rel_max(got, ref):
    got, ref = to_np(got).astype(np.float64), to_np(ref).astype(np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def rel_rms(got, ref):
    got, ref = to_np(got).astype(np.float64), to_np(ref).astype(np.float64)
    return float(np.sqrt(((got - ref) ** 2).mean()) / (np.sqrt((ref ** 2).mean()) + 1e-30))


def assert_close(got, ref, tol, what=""):
    assert to_np(got).shape == to_np(ref).shape, (what, to_np(got).shape, to_np(ref).shape)
    assert np.isfinite(to_np(got)).all(), what + ": non-finite output"
    e = rel_max(got, ref)
    assert e <= tol, "%s: rel max err %.3e > %.1e" % (what, e, tol)


def assert_close_bf16(got, ref, what=""):
    assert np.isfinite(to_np(got)).all(), what + ": non-finite output"
    e, r = rel_max(got, ref), rel_rms(got, ref)
    assert e <= TOL_BF16_MAX and r <= TOL_BF16_RMS, "%s: bf16 rel max %.3e rms %.3e" % (what, e, r)
