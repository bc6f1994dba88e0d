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


def _header_struct_fields(name):
    """[(ctype, field)] of `typedef struct name {...}` in include/hdrsky.h (comments stripped)."""
    text = open(os.path.join(ROOT, "include", "hdrsky.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*(?:\s*,\s*\**\s*[A-Za-z_][A-Za-z0-9_]*)*)$", decl)
        ctype, names = m.group(1).strip(), [n.strip() for n in m.group(2).split(",")]
        for n in names:
            out.append((ctype + ("*" if n.startswith("*") else ""), n.lstrip("* ")))
    return out


def test_struct_layouts_match_the_library_and_the_header():
    """The layout contract (hdrsky_sizeof / hdrsky_abi_version): every ctypes mirror has the library's size, the field
    names and order of the header, and load() refuses a mismatching mirror."""
    L = pkg("_lib")
    lib = L.load()
    assert lib.hdrsky_abi_version() == L.ABI_VERSION
    assert lib.hdrsky_sizeof(b"no_such_struct") == 0
    for cname, mirror in L.STRUCTS.items():
        assert lib.hdrsky_sizeof(cname.encode()) == ctypes.sizeof(mirror), cname
        hdr = _header_struct_fields(cname)
        assert [n for _, n in hdr] == [n for n, _ in mirror._fields_], cname
        for (ctype, n), (_, ft) in zip(hdr, mirror._fields_):
            want = (ctypes.c_void_p if "*" in ctype else {"int": ctypes.c_int32, "int32_t": ctypes.c_int32, "float": ctypes.c_float,
                                                          "hdrsky_conv_desc": L.ConvDesc}[ctype.replace("const ", "").strip()])
            assert ft is want, (cname, n, ctype, ft)
    assert ctypes.sizeof(L.ConvDesc) == 29 * 4


def test_integration_md_binding_stub_matches_the_built_library(monkeypatch):
    """INTEGRATION.md section 2 is what a maintainer copies: its code block is evaluated here (CPU only, nothing is
    launched) against the built library - its own size assertions run, and hdrsky_conv_desc_init must stay inside
    the structure it declares (round 2's stub was 16 bytes short)."""
    L = pkg("_lib")
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# hdrsky_binding\.py.*?)```", text, flags=re.S).group(1)
    monkeypatch.setenv("HDRSKY_LIB", L.LIB_PATH)
    ns = {}
    exec(compile(block, "INTEGRATION.md#2", "exec"), ns)
    stub = ns["ConvDesc"]
    assert ctypes.sizeof(stub) == ctypes.sizeof(L.ConvDesc) == ns["lib"].hdrsky_sizeof(b"hdrsky_conv_desc")
    assert [n for n, _ in stub._fields_] == [n for n, _ in L.ConvDesc._fields_]
    # a guard word right behind the structure survives the init call
    class Guarded(ctypes.Structure):
        _fields_ = [("d", stub), ("guard", ctypes.c_uint32 * 8)]
    g = Guarded()
    for i in range(8):
        g.guard[i] = 0xA5A5A5A5
    assert ns["lib"].hdrsky_conv_desc_init(ctypes.cast(ctypes.pointer(g), ctypes.POINTER(stub)), 2, 32, 128, 3, 32, 7, 7, 1, 1, 1) == 0
    assert all(v == 0xA5A5A5A5 for v in g.guard) and (g.d.Ho, g.d.Wo, g.d.pad_t) == (32, 128, 3)


def _plan_bytes(L, lib, jobs):
    """hdrsky_conv2d_wgrad_ws_bytes for a list of (B, H, W, Cin, Cout, k, stride, same, x_bf16, dy_bf16) - host-only planning:
    the pointers are never dereferenced (placeholders), nothing is launched."""
    arr = (L.WgradJob * len(jobs))()
    for j, (B, H, W, Cin, Cout, k, s, same, xb, yb) in zip(arr, jobs):
        assert lib.hdrsky_conv_desc_init(j.desc, B, H, W, Cin, Cout, k, k, s, int(same), 1) == 0
        j.desc.compute = L.HDRSKY_BF16
        j.x, j.dy, j.dw, j.db = 0x1000, 0x2000, 0x3000, 0x4000
        j.x_bf16, j.dy_bf16 = int(xb), int(yb)
    return int(lib.hdrsky_conv2d_wgrad_ws_bytes(arr, len(jobs)))


def test_weight_gradient_workspace_planning_is_host_only_and_matches_the_decompositions():
    """The deterministic weight-gradient path is planned on the host (hdrsky_conv2d_wgrad_ws_bytes: no launch, no GPU): the
    scratch of the three kernels' partial slabs for the layer classes of the training step (batch 32).
    conv_wgrad3_kernel (narrow side): 256 workgroups x [taps][CP or 64][wide block] slabs + bias partials;
    conv_wgrad2_kernel: a layer that is not split over pixels needs no slab at all; unsupported geometry: 0."""
    L = pkg("_lib")
    lib = L.load()
    f4 = 4
    # 7x7 3->32 stem @32x128: 1024 tiles of 128 pixels / 4 per workgroup = 256 chunks x (49 taps x 4 x 32) + 256 x 32 bias words
    stem = (32, 32, 128, 3, 32, 7, 1, True, 0, 1)
    assert _plan_bytes(L, lib, [stem]) == (256 * 49 * 4 * 32 + 256 * 32) * f4
    # 7x7 32->3 tail: the same decomposition with the roles swapped: slabs [49][32][4], bias partials 4 per chunk
    tail = (32, 32, 128, 32, 3, 7, 1, True, 1, 0)
    assert _plan_bytes(L, lib, [tail]) == (256 * 49 * 32 * 4 + 256 * 4) * f4
    # both in one call: one launch, the slabs side by side
    assert _plan_bytes(L, lib, [stem, tail]) == _plan_bytes(L, lib, [stem]) + _plan_bytes(L, lib, [tail])
    # 4x4 stride-2 6->64 first layer @32x128 (64-pixel tiles, >= 256 pixels per workgroup: 128 chunks), 8 padded channels
    d1 = (32, 32, 128, 6, 64, 4, 2, True, 0, 1)
    assert _plan_bytes(L, lib, [d1]) == (128 * 16 * 8 * 64 + 128 * 64) * f4
    # a wide layer with two final bf16 operands (LDS-DMA kernel) in a call of its own is split over pixels ...
    res = (32, 8, 32, 128, 128, 3, 1, True, 1, 1)
    one = _plan_bytes(L, lib, [res])
    assert one > 9 * 128 * 128 * f4 and one % f4 == 0
    # ... and shares the 256-workgroup budget by work in a call of twelve: fewer chunks per layer, less scratch per layer
    assert _plan_bytes(L, lib, [res] * 12) < 12 * one
    # a geometry no kernel takes (dilated conv): the planner says so with 0
    arr = (L.WgradJob * 1)()
    assert lib.hdrsky_conv_desc_init(arr[0].desc, 2, 8, 32, 32, 32, 3, 3, 1, 1, 1) == 0
    arr[0].desc.dilate = 2
    arr[0].x, arr[0].dy, arr[0].dw = 0x1000, 0x2000, 0x3000
    assert int(lib.hdrsky_conv2d_wgrad_ws_bytes(arr, 1)) == 0


def test_no_packed_f32_with_src1_op_sel(tmp_path):
    """The built library holds no VOP3P packed-f32 instruction whose LOW result reads src1's HIGH dword (op_sel[1] = 1):
    on MI355X that form returns a wrong low half in lanes 48-63 while another wave of the compute unit issues MFMAs back to
    back (profiles/experiments/pk_hazard/pk_opsel.hip, profiles/r04_pk_opsel_erratum.txt) - the root cause of the
    irreproducible distortion-aware data gradient of round 3.  The build avoids it with
    -fno-slp-vectorize (csrc/Makefile); this test disassembles what was built, so a flag lost in a refactoring, a new
    translation unit with its own flags or explicit <2 x float> arithmetic cannot bring the form back unnoticed."""
    import shutil
    import subprocess
    import pytest
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    L = pkg("_lib")
    lib = str(tmp_path / "libhdrsky.so")
    shutil.copy(L.LIB_PATH, lib)
    subprocess.run([objdump, "--offloading", lib], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert objs, "no gfx950 code object extracted from the library"
    # every VOP3P packed-f32 arithmetic form (mul / fma / add were measured; min / max and whatever a later ISA revision adds
    # are treated alike - v_pk_mov_b32 selects by op_sel alone and is the one packed move the erratum does not touch)
    pat = re.compile(r"\b(v_pk_[a-z0-9]+_f32)\b(.*)")
    mk = open(os.path.join(os.path.dirname(L.LIB_PATH), "csrc", "Makefile")).read()
    flags = [ln for ln in mk.splitlines() if ln.startswith("CXXFLAGS")]
    assert flags and all("-fno-slp-vectorize" in ln for ln in flags), "csrc/Makefile: -fno-slp-vectorize must be in CXXFLAGS"
    total, bad, mfma = 0, [], 0
    for f in objs:
        text = subprocess.run([objdump, "-d", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        mfma += text.count("v_mfma_")
        for m in pat.finditer(text):
            total += 1
            sel = re.search(r"op_sel:\[([01](?:,[01])*)\]", m.group(2))
            if sel and len(sel.group(1).split(",")) >= 2 and sel.group(1).split(",")[1] == "1":
                bad.append(m.group(0).split("//")[0].strip())
    assert mfma > 1000, "the disassembly does not look like the library's device code"
    assert not bad, "%d packed-f32 instructions with op_sel[src1] = 1, e.g. %s" % (len(bad), bad[:3])
    assert total < 64, "%d packed-f32 instructions: is -fno-slp-vectorize still in csrc/Makefile?" % total


def test_tuning_hooks_are_ignored_without_the_experiments_gate(monkeypatch):
    """csrc/hooks.h / <pkg>/hooks.py: switches are always honoured, tuning hooks only under HDRSKY_EXPERIMENTS=1 - a stray
    HDRSKY_TILE in a deployment's environment must not change which kernel the product launches.  Checked on the host side of
    the library (hdrsky_conv_kernel_name needs no GPU) and on the Python side."""
    L, HK = pkg("_lib"), pkg("hooks")
    lib = L.load()
    d = L.ConvDesc()
    assert lib.hdrsky_conv_desc_init(d, 32, 16, 64, 64, 64, 3, 3, 1, 1, 1) == 0
    d.compute = L.BF16 if hasattr(L, "BF16") else 0

    def name():
        buf = ctypes.create_string_buffer(256)
        assert lib.hdrsky_conv_kernel_name(d, buf, 256) == 0
        return buf.value.decode()
    import ctypes
    base = name()
    monkeypatch.setenv("HDRSKY_TILE", "2,2,2,2,32,0")
    monkeypatch.setenv("HDRSKY_RAW_BF16", "1")
    assert name() == base and not lib.hdrsky_experiments_enabled() and not HK.H.raw_bf16 and not HK.H.experiments
    monkeypatch.setenv("HDRSKY_EXPERIMENTS", "1")
    assert lib.hdrsky_experiments_enabled() and HK.H.raw_bf16 and "<2, 2, 2, 2, 32" in name() and name() != base
    monkeypatch.delenv("HDRSKY_EXPERIMENTS")
    assert name() == base and not HK.H.raw_bf16
    # a switch is honoured without the gate
    monkeypatch.setenv("HDRSKY_DA_MAT", "0")
    assert not HK.H.da_mat
    monkeypatch.delenv("HDRSKY_DA_MAT")
    assert HK.H.da_mat
