"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/hdrsky.h declares, and its host-side
helpers agree with the TF padding rules.  No kernel is launched."""
import ctypes
import os
import re

from conftest import ROOT, pkg
from oracle import tfsem as T


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hdrsky.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hdrsky_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = pkg("_lib")
    lib = L.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    # and the Python binding table covers the header one-to-one
    assert sorted(L.SIGNATURES) == syms
    assert lib.hdrsky_version().startswith(b"hdrsky")


def test_conv_desc_matches_tf_padding():
    L = pkg("_lib")
    lib = L.load()
    for (H, W, k, s, same) in [(32, 128, 7, 1, 1), (32, 128, 3, 2, 1), (16, 64, 4, 2, 1), (4, 16, 4, 1, 1), (4, 16, 4, 1, 0),
                               (9, 37, 3, 2, 1), (10, 20, 3, 1, 1)]:
        d = L.ConvDesc()
        assert lib.hdrsky_conv_desc_init(d, 2, H, W, 32, 64, k, k, s, same, 1) == 0
        if same:
            assert (d.Ho, d.Wo) == (-(-H // s), -(-W // s))
            assert d.pad_t == T.same_pad(H, k, s)[0] and d.pad_l == T.same_pad(W, k, s)[0]
        else:
            assert (d.Ho, d.Wo) == ((H - k) // s + 1, (W - k) // s + 1) and d.pad_t == 0
    d = L.ConvDesc()
    assert lib.hdrsky_conv_desc_init(d, 2, 8, 32, 128, 64, 3, 3, 1, 1, 2) == 0 and (d.Ho, d.Wo, d.Hc, d.Wc) == (16, 64, 16, 64)
    assert lib.hdrsky_conv_desc_init(d, 2, 8, 32, 128, 64, 3, 3, 3, 1, 1) == L.HDRSKY_BF16 - 1  # stride 3 -> EINVAL
    assert lib.hdrsky_fc_nsplit(8192) == 4 and lib.hdrsky_fc_nsplit(512) == 2
    assert lib.hdrsky_conv_packed_elems(3, 3, 128, 128) == (9 * 4 + 1) * 4 * 128 * 8


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    L = pkg("_lib")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        L.load()
    except RuntimeError as e:
        assert "no CPU/PyTorch fallback" in str(e)
    else:
        raise AssertionError("load() must raise when libhdrsky.so is absent")
