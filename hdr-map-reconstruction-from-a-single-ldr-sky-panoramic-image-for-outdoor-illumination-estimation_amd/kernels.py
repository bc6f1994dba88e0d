"""Typed torch-tensor front end of the C ABI (one thin function per libhdrsky entry point).

PyTorch is plumbing here: device allocations (``torch.empty``), the current HIP stream and
``data_ptr()``; every FLOP happens inside libhdrsky.so.  All functions validate shapes / dtypes /
contiguity on the host before a kernel is enqueued (a mis-shaped operand must never reach the GPU).
"""
import contextlib
import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from . import hooks as HOOKS

BF16, BF16X3 = L.HDRSKY_BF16, L.HDRSKY_BF16X3
IN_EPS = 1e-3  # tfa InstanceNormalization / Keras BatchNormalization default epsilon


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- launch trace (bench.py's per-layer roofline rows) -----------------------------------------------------------------
# TRACE = [] switches it on: every matrix-core conv / weight-gradient / sample-resident launch appends
# dict(kind, label, kernel, shape, flop, relaunch): `flop` = ALGORITHMIC 2*MAC of the layer (a zero-stuffed stride-2 data
# gradient counts the forward conv's work, not the stuffed zeros), `relaunch()` re-enqueues the identical launch on the
# current stream (it keeps its operands alive).  label(name) names the next traced launch(es) - the trainer sets it.
TRACE = None
_LABEL = [None]
WG_NAMES = {}          # data_ptr of a weight-gradient destination -> layer name (set by the trainer; labels of traced calls)


def label(name):
    _LABEL[0] = name


def _trace(kind, kernel, shape, flop, relaunch, default_label=None):
    if TRACE is not None:
        TRACE.append(dict(kind=kind, label=_LABEL[0] or default_label, kernel=kernel, shape=shape, flop=float(flop), relaunch=relaunch))
        _LABEL[0] = None          # a label names ONE traced launch (a stale one mislabelled the res-chain rows in round 3)


@contextlib.contextmanager
def no_gc():
    """For hipGraph captures: collects garbage NOW and keeps the cyclic collector off inside the block.  torch.cuda.graph no
    longer collects on entry (torch >= 2.9: only with torch.compiler.config.force_cudagraph_gc), so an unreachable object that
    owns a hipGraphExec or an event - a Trainer of an earlier test, kept alive by its closures' reference cycle - could be
    finalised by a collection that happens to start in the middle of a capture: hipGraphExecDestroy on a capturing thread
    aborts the process (seen once in the GPU suite, in the capture of test_split_discriminator_passes_equal_the_paired_batch)."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


_HIP = [None]


def masked_stream(cu_lo, cu_hi, ncu=256, step=1):
    """A HIP stream whose kernels run on compute units [cu_lo, cu_hi) only (hipExtStreamCreateWithCUMask on the HIP runtime torch
    loaded), wrapped as a torch stream.  Experiments of profiles/cu_mask_*.py: reserving compute units for the dependent chain of
    small launches that the step / the forward pass waits for."""
    if _HIP[0] is None:
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("masked_stream: the HIP runtime is not loaded yet (touch the GPU first)")
        _HIP[0] = ctypes.CDLL(path)
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(cu_lo, cu_hi, step):      # (step > 1: every step-th unit of the range - a mask spread over the dies)
        mask[cu // 32] |= 1 << (cu % 32)
    st = ctypes.c_void_p()
    rc = _HIP[0].hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(st.value)


def conv_kernel_name(d):
    buf = ctypes.create_string_buffer(160)
    L.check(L.load().hdrsky_conv_kernel_name(d, buf, 160), "conv_kernel_name")
    return buf.value.decode()


def _p(t):
    return None if t is None else t.data_ptr()


def _raw(x):
    """Validates a raw conv output - fp32, or bfloat16 as the single-product mode stores it in front of a norm layer -
    and returns 1 for bfloat16 storage (the x_bf16 argument / flag bit of its readers)."""
    if torch.is_tensor(x) and x.dtype == torch.bfloat16:
        _bf16(x)
        return 1
    _f32(x)
    return 0


def _f32(t, *shape):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("expected a contiguous CUDA float32 tensor")
    if shape and tuple(t.shape) != tuple(shape):
        raise ValueError("shape %s != expected %s" % (tuple(t.shape), tuple(shape)))
    return t


@dataclass
class Stats:
    """Per-tile (sum, sumsq) partials [B][nparts][2][C] a conv emitted for the next normalisation."""
    part: torch.Tensor
    nparts: int
    count: int  # H*W the sums run over


@dataclass
class InXf:
    """Fused operand transform of a conv: act(norm(x)).  mode: IN_NONE / IN_AFFINE / IN_PARTIALS."""
    mode: int = L.IN_NONE
    slope: float = 1.0
    scale: Optional[torch.Tensor] = None   # AFFINE [B,C] or [C]
    shift: Optional[torch.Tensor] = None
    stats: Optional[Stats] = None          # PARTIALS
    gamma: Optional[torch.Tensor] = None
    beta: Optional[torch.Tensor] = None
    eps: float = IN_EPS
    gamma2: Optional[torch.Tensor] = None  # PARTIALS of a PAIRED tensor (two layers' outputs as one batch): the second half's layer
    beta2: Optional[torch.Tensor] = None


@dataclass
class ConvPair:
    """Second parameter set of a PAIRED conv launch (hdrsky_conv2d_fwd_pair): samples [B/2, B) of the launch run on this filter /
    bias / residual (and on InXf.gamma2 / beta2 of a PARTIALS transform); samples [0, B/2) on conv2d's own arguments."""
    pw: "PackedConv"
    bias: Optional[torch.Tensor] = None
    residual: Optional[torch.Tensor] = None


class Operand:
    """Explicit handle of a bf16 operand that a FORWARD launch wrote for the weight gradient of its layer - the caller owns it and
    hands the same object to the forward call and to the weight-gradient job (rounds 3-4 parked these tensors as attributes on
    the input tensor object, keyed by nothing that says whether the input was rewritten in between).
      conv2d(..., emit_xb=op)     -> op.tensor = act(norm(x)) as bf16 (hdrsky_conv2d_fwd_emit), op.key = the transform object
      da_conv2d(..., operand=op)  -> op.tensor = the gathered operand G [B,H,W,k*k*C], op.key = (offsets, k, input pointer, shape)
      wgrad_job / da_wgrad_job(..., operand=op) use op.tensor when op.key matches their arguments and fall back to x otherwise.
    The operand is what the forward pass multiplied: a gradient through it is the gradient of THAT forward, whatever happened
    to x since.  A forward call without a handle keeps nothing (an inference pass does not hold 4.5x its input)."""
    __slots__ = ("tensor", "key")

    def __init__(self):
        self.tensor, self.key = None, None

    def clear(self):
        self.tensor, self.key = None, None


class PackedConv:
    """MFMA B-operand image of a [KH,KW,Cin,Cout] filter (bf16 hi plane + optional lo residual plane).
    transpose_flip=True packs the data-gradient filter w'[ky,kx,co,ci] = w[KH-1-ky,KW-1-kx,ci,co]."""

    def __init__(self, w, precise=True, transpose_flip=False):
        _f32(w)
        kh, kw, ci, co = w.shape
        self.KH, self.KW = kh, kw
        self.Cin, self.Cout = (co, ci) if transpose_flip else (ci, co)
        self.flip = transpose_flip
        lib = L.load()
        n = lib.hdrsky_conv_packed_elems(kh, kw, self.Cin, self.Cout)
        self.hi = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        self.lo = torch.empty(n, dtype=torch.bfloat16, device=w.device) if precise else None
        self.repack(w)

    def repack(self, w):
        lib = L.load()
        L.check(lib.hdrsky_conv_pack_weights(_p(w), self.KH, self.KW, self.Cin, self.Cout, int(self.flip),
                                             _p(self.hi), _p(self.lo), _stream()), "conv_pack_weights")

    def as_1x1(self):
        """The same image read as the filter of a 1x1 convolution over KH*KW*Cin channels (its k-steps are in exactly that
        order: tap-major, 32-channel blocks) - the matmul of a distortion-aware layer on its gathered operand."""
        v = object.__new__(PackedConv)
        v.KH = v.KW = 1
        v.Cin, v.Cout, v.flip, v.hi, v.lo = self.KH * self.KW * self.Cin, self.Cout, self.flip, self.hi, self.lo
        return v


class MultiPacker:
    """One-launch re-pack of a list of (fp32 HWIO weight view, PackedConv) pairs (hdrsky_conv_pack_weights_multi)."""

    def __init__(self, pairs):
        rows, blk = [], 0
        for w, pw in pairs:
            n = pw.hi.numel()
            rows.append([w.data_ptr(), pw.hi.data_ptr(), pw.lo.data_ptr() if pw.lo is not None else 0, pw.KH, pw.KW,
                         pw.Cin, pw.Cout, int(pw.flip), blk])
            blk += (n + 2047) // 2048
        self.njobs, self.blocks = len(rows), blk
        self.table = torch.tensor(rows, dtype=torch.int64, device=pairs[0][0].device)
        self._keep = pairs   # the pointers in the table must stay valid

    def run(self):
        L.check(L.load().hdrsky_conv_pack_weights_multi(_p(self.table), self.njobs, self.blocks, _stream()),
                "conv_pack_weights_multi")


def conv_desc(B, H, W, Cin, Cout, KH, KW, stride=1, same=True, upsample=1):
    d = L.ConvDesc()
    L.check(L.load().hdrsky_conv_desc_init(d, B, H, W, Cin, Cout, KH, KW, stride, int(same), upsample), "conv_desc_init")
    return d


def _bf16(t, *shape):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise ValueError("expected a contiguous CUDA bfloat16 tensor")
    if shape and tuple(t.shape) != tuple(shape):
        raise ValueError("shape %s != expected %s" % (tuple(t.shape), tuple(shape)))
    return t


def conv2d(x, pw: PackedConv, bias=None, stride=1, same=True, upsample=1, xf: Optional[InXf] = None,
           out_slope=1.0, residual=None, final_relu=False, want_stats=False, compute=BF16, desc=None, out=None,
           out_bf16=False, mask_bf16=None, mask_slope=0.0, emit_xb: Optional["Operand"] = None, pair: Optional["ConvPair"] = None,
           x_shared=False):
    """y = final_relu(act(conv(xf(x)) + bias) + residual); returns (y, Stats|None).
    HDRSKY_BF16 mode: x may be a bfloat16 tensor (a final activation: no xf), out_bf16 stores y as bfloat16; mask_bf16
    (instead of residual): a bfloat16 ACTIVATED tensor of y's shape - y is multiplied by (it > 0 ? 1 : mask_slope), the
    activation backward fused into a data-gradient conv."""
    lib = L.load()
    if pair is not None:
        return _conv2d_pair(x, pw, bias, stride, same, upsample, xf, out_slope, residual, final_relu, want_stats, compute, desc, out,
                            out_bf16, mask_bf16, mask_slope, pair, x_shared)
    if torch.is_tensor(x) and x.dtype == torch.bfloat16:
        _bf16(x)
    else:
        _f32(x)
    B, H, W, C = x.shape
    if C != pw.Cin:
        raise ValueError("Cin mismatch: x has %d, filter %d" % (C, pw.Cin))
    d = desc if desc is not None else conv_desc(B, H, W, C, pw.Cout, pw.KH, pw.KW, stride, same, upsample)
    if (d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW) != (B, H, W, C, pw.Cout, pw.KH, pw.KW):
        raise ValueError("descriptor does not match operands")
    d.compute = compute
    if compute == BF16X3 and pw.lo is None:
        raise ValueError("BF16X3 needs the lo weight plane (PackedConv(precise=True))")
    xf_given = xf
    xf = xf or InXf()
    d.in_mode, d.in_slope = xf.mode, float(xf.slope)
    in_scale = in_shift = in_part = in_gamma = in_beta = None
    if xf.mode == L.IN_AFFINE:
        in_scale, in_shift = _f32(xf.scale), _f32(xf.shift)
        if in_scale.numel() == B * C:
            d.ss_bstride = C
        elif in_scale.numel() == C:
            d.ss_bstride = 0
        else:
            raise ValueError("affine table must have C or B*C entries")
        if in_shift.numel() != in_scale.numel():
            raise ValueError("scale/shift size mismatch")
    elif xf.mode == L.IN_PARTIALS:
        st = xf.stats
        in_part = _f32(st.part, B, st.nparts, 2, C)
        if st.count != H * W:
            raise ValueError("partials were not accumulated over this tensor's H*W")
        in_gamma, in_beta = _f32(xf.gamma, C), _f32(xf.beta, C)
        d.in_nparts, d.in_eps = st.nparts, float(xf.eps)
    d.out_slope, d.final_relu, d.want_stats = float(out_slope), int(bool(final_relu)), int(bool(want_stats))
    d.x_bf16, d.y_bf16 = int(x.dtype == torch.bfloat16), int(bool(out_bf16))
    if bias is not None:
        _f32(bias, pw.Cout)
    ydt = torch.bfloat16 if out_bf16 else torch.float32
    y = out if out is not None else torch.empty((B, d.Ho, d.Wo, pw.Cout), dtype=ydt, device=x.device)
    (_bf16 if out_bf16 else _f32)(y, B, d.Ho, d.Wo, pw.Cout)
    if mask_bf16 is not None:
        if residual is not None:
            raise ValueError("residual and mask_bf16 exclude each other")
        residual = _bf16(mask_bf16, B, d.Ho, d.Wo, pw.Cout)
        d.res_mode, d.mask_slope = 1, float(mask_slope)
    elif residual is not None:
        _f32(residual, B, d.Ho, d.Wo, pw.Cout)
    stats = None
    if want_stats:
        nparts = lib.hdrsky_conv_stats_nparts(d)
        stats = Stats(torch.empty((B, nparts, 2, pw.Cout), dtype=torch.float32, device=x.device), nparts, d.Ho * d.Wo)
    args = (d, _p(x), _p(pw.hi), _p(pw.lo), _p(bias), _p(in_scale), _p(in_shift), _p(in_part), _p(in_gamma), _p(in_beta),
            _p(residual), _p(y), _p(stats.part) if stats else None)
    # emit_xb (an Operand handle): the transformed operand act(norm(x)) written as a bf16 tensor by this launch - the x of the
    # layer's weight gradient, handed over through the handle together with the transform it belongs to (wgrad_job(operand=))
    if emit_xb is not None:
        emit_xb.clear()
    if emit_xb is not None and xf_given is not None and lib.hdrsky_conv2d_emit_supported(d):
        xb = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=x.device)
        L.check(lib.hdrsky_conv2d_fwd_emit(*args, _p(xb), _stream()), "conv2d_fwd_emit")
        emit_xb.tensor, emit_xb.key = xb, xf_given
    else:
        L.check(lib.hdrsky_conv2d_fwd(*args, _stream()), "conv2d_fwd")
    if TRACE is not None:
        keep = (x, pw, bias, in_scale, in_shift, in_part, in_gamma, in_beta, residual, y, stats)
        stuffed = 4 if d.dilate == 2 else 1
        kname = conv_kernel_name(d)
        # a stride-2 data gradient: by output phases (the PH instantiations, last template argument) or on the zero-stuffed operand
        form = " up2" if d.upsample == 2 else ((" s2 dgrad by phases" if kname.endswith("true, false>") else " s2 dgrad zero-stuffed") if stuffed == 4 else "")
        _trace("dgrad" if pw.flip else "conv", kname,
               "%dx%d %d->%d @%dx%d B=%d%s%s" % (d.KH, d.KW, C, pw.Cout, d.Ho, d.Wo, B, " s2" if d.stride == 2 else "", form),
               2.0 * B * d.Ho * d.Wo * d.KH * d.KW * C * pw.Cout / stuffed,
               lambda a_=args, k_=keep: L.check(lib.hdrsky_conv2d_fwd(*a_, _stream()), "conv2d_fwd"))
    return y, stats


def _half_xf(xf, lo, hi, second):
    """The operand transform of one half of a paired tensor (the fallback of _conv2d_pair: two separate launches)."""
    if xf is None or xf.mode == L.IN_NONE:
        return xf
    if xf.mode == L.IN_AFFINE:
        if xf.scale.dim() == 1:
            return xf
        return InXf(mode=L.IN_AFFINE, slope=xf.slope, scale=xf.scale[lo:hi], shift=xf.shift[lo:hi])
    st = xf.stats
    return InXf(mode=L.IN_PARTIALS, slope=xf.slope, stats=Stats(st.part[lo:hi], st.nparts, st.count), eps=xf.eps,
                gamma=xf.gamma2 if second else xf.gamma, beta=xf.beta2 if second else xf.beta)


def _conv2d_pair(x, pw, bias, stride, same, upsample, xf, out_slope, residual, final_relu, want_stats, compute, desc, out, out_bf16,
                 mask_bf16, mask_slope, pair, x_shared):
    """conv2d(..., pair=ConvPair): two layers of identical geometry as ONE launch (hdrsky_conv2d_fwd_pair) on a batch of 2 Bh samples -
    the first Bh on (pw, bias, residual, xf.gamma / beta), the last Bh on the pair's.  x: [2 Bh, ...], or [Bh, ...] with x_shared (both
    layers read the same input).  Returns (y [2 Bh, ...], Stats of 2 Bh samples): bit-identical to the two separate launches, which
    is also what runs where the library has no paired instantiation for the layer's tile."""
    lib = L.load()
    (_bf16 if x.dtype == torch.bfloat16 else _f32)(x)
    Bx, H, W, C = x.shape
    Bt = 2 * Bx if x_shared else Bx
    pw2 = pair.pw
    if Bt % 2 or C != pw.Cin or (pw2.KH, pw2.KW, pw2.Cin, pw2.Cout, pw2.flip) != (pw.KH, pw.KW, pw.Cin, pw.Cout, pw.flip):
        raise ValueError("conv2d pair: two filters of one geometry on an even batch")
    Bh = Bt // 2
    if desc is not None:
        d = L.ConvDesc(); ctypes.memmove(ctypes.byref(d), ctypes.byref(desc), ctypes.sizeof(d))
        if d.B != Bt:
            raise ValueError("conv2d pair: the descriptor must describe the whole batch")
    else:
        d = conv_desc(Bt, H, W, C, pw.Cout, pw.KH, pw.KW, stride, same, upsample)
    d.compute = compute
    xfo = xf or InXf()
    if x_shared and xfo.mode != L.IN_NONE:
        raise ValueError("conv2d pair: a shared input takes no operand transform")
    dh = L.ConvDesc(); ctypes.memmove(ctypes.byref(dh), ctypes.byref(d), ctypes.sizeof(d)); dh.B = Bh     # one layer's launch
    tabs = _xf_args(d, xf, Bt, H, W, C)
    dh.in_mode, dh.in_slope, dh.ss_bstride, dh.in_nparts, dh.in_eps = d.in_mode, d.in_slope, d.ss_bstride, d.in_nparts, d.in_eps
    g2 = b2 = None
    if xfo.mode == L.IN_PARTIALS:
        g2, b2 = _f32(xfo.gamma2, C), _f32(xfo.beta2, C)
    for dd in (d, dh):
        dd.out_slope, dd.final_relu, dd.want_stats = float(out_slope), int(bool(final_relu)), int(bool(want_stats))
        dd.x_bf16, dd.y_bf16 = int(x.dtype == torch.bfloat16), int(bool(out_bf16))
    if (bias is None) != (pair.bias is None):
        raise ValueError("conv2d pair: both layers with or without a bias")
    if bias is not None:
        _f32(bias, pw.Cout); _f32(pair.bias, pw.Cout)
    ydt = torch.bfloat16 if out_bf16 else torch.float32
    y = out if out is not None else torch.empty((Bt, d.Ho, d.Wo, pw.Cout), dtype=ydt, device=x.device)
    (_bf16 if out_bf16 else _f32)(y, Bt, d.Ho, d.Wo, pw.Cout)
    res1, res2 = residual, pair.residual
    if mask_bf16 is not None:
        raise ValueError("conv2d pair: no activation-mask form")
    if (res1 is None) != (res2 is None):
        raise ValueError("conv2d pair: both layers with or without a residual")
    if res1 is not None:
        _f32(res1, Bh, d.Ho, d.Wo, pw.Cout); _f32(res2, Bh, d.Ho, d.Wo, pw.Cout)
    stats = None
    if want_stats:
        nparts = lib.hdrsky_conv_stats_nparts(dh)
        stats = Stats(torch.empty((Bt, nparts, 2, pw.Cout), dtype=torch.float32, device=x.device), nparts, d.Ho * d.Wo)
    in_scale, in_shift, in_part, in_gamma, in_beta = tabs
    rc = lib.hdrsky_conv2d_fwd_pair(d, _p(x), int(bool(x_shared)), _p(pw.hi), _p(pw.lo), _p(bias), _p(pw2.hi), _p(pw2.lo), _p(pair.bias),
                                    _p(in_scale), _p(in_shift), _p(in_part), _p(in_gamma), _p(in_beta), _p(g2), _p(b2), _p(res1), _p(res2),
                                    _p(y), _p(stats.part) if stats else None, _stream())
    if rc == L.HDRSKY_EUNSUPPORTED:      # no paired instantiation for this tile / mode: the two launches
        for hf, (pwh, bh, rh) in enumerate(((pw, bias, res1), (pw2, pair.bias, res2))):
            lo, hi = hf * Bh, (hf + 1) * Bh
            dsc = None
            if desc is not None:
                dsc = L.ConvDesc(); ctypes.memmove(ctypes.byref(dsc), ctypes.byref(desc), ctypes.sizeof(dsc)); dsc.B = Bh
            yh, sth = conv2d(x if x_shared else x[lo:hi], pwh, bh, stride=stride, same=same, upsample=upsample, xf=_half_xf(xf, lo, hi, hf == 1),
                             out_slope=out_slope, residual=rh, final_relu=final_relu, want_stats=want_stats, compute=compute, desc=dsc,
                             out=y[lo:hi], out_bf16=out_bf16)
            if stats is not None:
                stats.part[lo:hi].copy_(sth.part)
        return y, stats
    L.check(rc, "conv2d_fwd_pair")
    if TRACE is not None:
        args = (d, _p(x), int(bool(x_shared)), _p(pw.hi), _p(pw.lo), _p(bias), _p(pw2.hi), _p(pw2.lo), _p(pair.bias), _p(in_scale), _p(in_shift),
                _p(in_part), _p(in_gamma), _p(in_beta), _p(g2), _p(b2), _p(res1), _p(res2), _p(y), _p(stats.part) if stats else None)
        keep = (x, pw, pw2, bias, pair.bias, in_scale, in_shift, in_part, in_gamma, in_beta, g2, b2, res1, res2, y, stats)
        stuffed = 4 if d.dilate == 2 else 1
        _trace("dgrad" if pw.flip else "conv", conv_kernel_name(dh).replace("false>", "false, true>") + " (paired)",
               "2 x %dx%d %d->%d @%dx%d B=%d%s" % (d.KH, d.KW, C, pw.Cout, d.Ho, d.Wo, Bh, " s2" if d.stride == 2 else ""),
               2.0 * Bt * d.Ho * d.Wo * d.KH * d.KW * C * pw.Cout / stuffed,
               lambda a_=args, k_=keep: L.check(lib.hdrsky_conv2d_fwd_pair(*a_, _stream()), "conv2d_fwd_pair"))
    return y, stats


def _xf_args(d, xf, B, H, W, C):
    """Fills the operand-transform fields of a descriptor; returns the five table pointers."""
    xf = xf or InXf()
    d.in_mode, d.in_slope = xf.mode, float(xf.slope)
    in_scale = in_shift = in_part = in_gamma = in_beta = None
    if xf.mode == L.IN_AFFINE:
        in_scale, in_shift = _f32(xf.scale), _f32(xf.shift)
        d.ss_bstride = C if in_scale.numel() == B * C else 0
        if in_scale.numel() not in (C, B * C) or in_shift.numel() != in_scale.numel():
            raise ValueError("affine table must have C or B*C entries")
    elif xf.mode == L.IN_PARTIALS:
        st = xf.stats
        in_part = _f32(st.part, B, st.nparts, 2, C)
        if st.count != H * W:
            raise ValueError("partials were not accumulated over this tensor's H*W")
        in_gamma, in_beta = _f32(xf.gamma, C), _f32(xf.beta, C)
        d.in_nparts, d.in_eps = st.nparts, float(xf.eps)
    return in_scale, in_shift, in_part, in_gamma, in_beta


def conv2d_wgrad(x, dy, KH, KW, stride=1, same=True, upsample=1, xf: Optional[InXf] = None, compute=BF16,
                 want_db=True, dw=None, db=None, deterministic=True):
    """(dw [KH,KW,Cin,Cout], db [Cout]) of the conv whose forward consumed xf(x) and produced dy's shape.
    Accumulates into dw/db when given (they must then already hold valid values), else allocates zeros."""
    _f32(x); _f32(dy)
    B, H, W, C = x.shape
    Cout = dy.shape[-1]
    d = conv_desc(B, H, W, C, Cout, KH, KW, stride, same, upsample)
    if tuple(dy.shape) != (B, d.Ho, d.Wo, Cout):
        raise ValueError("dy shape %s does not match the conv output %s" % (tuple(dy.shape), (B, d.Ho, d.Wo, Cout)))
    d.compute = compute
    tabs = _xf_args(d, xf, B, H, W, C)
    if dw is None:
        dw = torch.zeros((KH, KW, C, Cout), dtype=torch.float32, device=x.device)
    if db is None and want_db:
        db = torch.zeros((Cout,), dtype=torch.float32, device=x.device)
    _f32(dw, KH, KW, C, Cout)
    conv2d_wgrad_multi([(d, x, dy, tabs, dw, db)], deterministic=deterministic)
    return dw, db


def _da_key(x, offs, ksize):
    return (offs.data_ptr(), int(ksize), x.data_ptr(), tuple(x.shape), x.dtype)


def da_wgrad_job(x, dy, ksize, offs, dw, db=None, compute=BF16, operand: Optional["Operand"] = None):
    """conv2d_wgrad_multi entry for a distortion-aware layer (distortion_aware_ops.conv2d, kernel [k*k*C, F]): dw [k*k*C, F]
    += G^T dY with the gathered operand G recomputed inside the launch (never in memory), or - single-product mode from 1024
    pixels per sample (da_mat_ok) - a plain 1x1 weight gradient on the written operand: the one the forward left in `operand`
    (da_conv2d(operand=)), else gathered here from x.  x [B,H,W,C] fp32, C % 32 == 0."""
    B, H, W, C = x.shape
    F = dy.shape[-1]
    k2 = ksize * ksize
    if tuple(dy.shape) != (B, H, W, F) or C % 32 or tuple(offs.shape) != (H, k2, 2):
        raise ValueError("da_wgrad_job: x %s, dy %s, offs %s" % (tuple(x.shape), tuple(dy.shape), tuple(offs.shape)))
    if da_mat_ok(compute, ksize, C, H * W, "wgrad"):   # dW = G^T dY: a plain 1x1 weight gradient on the gathered operand (the forward's, or written here)
        # the operand the layer's forward wrote (operand: the caller's handle), else gathered here from x as it stands
        G = operand.tensor if operand is not None and operand.tensor is not None and operand.key == _da_key(x, offs, ksize) \
            else da_gather_bf16(x, offs, ksize=ksize)
        if dy.dtype != torch.bfloat16:      # (a gradient the fused data-gradient kernel reads as fp32: the LDS-DMA kernel wants bf16)
            dy = to_bf16(_f32(dy))
        _f32(dw, k2 * C, F)
        return wgrad_job(G, dy, 1, 1, dw.view(1, 1, k2 * C, F), db, compute=compute)
    _f32(x); _f32(dy); _f32(offs)
    d = conv_desc(B, H, W, k2 * C, F, 1, 1, 1, True, 1)
    d.compute = compute
    tabs = _xf_args(d, None, B, H, W, k2 * C)
    _f32(dw, k2 * C, F)
    if db is not None:
        _f32(db, F)
    return (d, x, dy, tabs, dw, db, (offs, ksize, C))


def _da_wgrad_region(job):
    """Kernel gradient of one distortion-aware layer through hdrsky_da_conv2d_wgrad (the tile group's source rows staged in
    LDS; BF16 mode, offsets from da_offsets_device).  False: not applicable here - the job stays in the generic launch."""
    d, x, dy, _, dw, db, (offs, ksize, C) = job
    rows = getattr(offs, "da_rows", None)
    if rows is None or d.compute != BF16 or C > HOOKS.H.da_wgrad_region_maxc:
        return False      # (wide layers: measured inside the training step, the grouped generic launch is no slower)
    B, H, W, _ = x.shape
    F = dy.shape[-1]
    lib = L.load()
    row_lo, spans = rows[0].data_ptr(), rows[1].ctypes.data
    nbytes = int(lib.hdrsky_da_conv2d_wgrad_ws_bytes(row_lo, spans, B, H, W, C, F, ksize))
    if nbytes <= 0:
        return False
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    L.check(lib.hdrsky_da_conv2d_wgrad(_p(x), _p(dy), int(dy.dtype == torch.bfloat16), _p(offs), row_lo, spans, B, H, W, C, F,
                                       ksize, _p(dw), _p(db), _p(ws), nbytes, _stream()), "da_conv2d_wgrad")
    return True


def wgrad_job(x, dy, KH, KW, dw, db=None, stride=1, same=True, upsample=1, xf: Optional[InXf] = None, compute=BF16,
              operand: Optional["Operand"] = None):
    """One entry for conv2d_wgrad_multi: dw [KH,KW,Cin,Cout] (+= ; must hold valid values, e.g. zeros), db [Cout] or None.
    x / dy may be bf16 tensors: final activations / gradients (no operand transform), or - x with a transform - a raw conv
    output as the single-product mode stores it (materialised by conv2d_wgrad_multi, or widened by the narrow-output kernel).
    operand: the handle the layer's forward call filled (conv2d(emit_xb=op)) - used when it belongs to this transform.
    The returned job references its tensors, which keeps them alive until the launch."""
    if operand is not None and operand.tensor is not None and xf is not None and operand.key is xf and upsample == 1 and \
            compute == BF16 and dy.dtype == torch.bfloat16 and tuple(operand.tensor.shape) == tuple(x.shape):
        x, xf = operand.tensor, None        # the forward launch wrote act(norm(x)) as bf16 (conv2d(emit_xb=op)): the final operand
    for t in (x, dy):
        if not (torch.is_tensor(t) and t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)):
            raise ValueError("expected contiguous CUDA float32 / bfloat16 tensors")
    B, H, W, C = x.shape
    Cout = dy.shape[-1]
    d = conv_desc(B, H, W, C, Cout, KH, KW, stride, same, upsample)
    if tuple(dy.shape) != (B, d.Ho, d.Wo, Cout):
        raise ValueError("dy shape %s does not match the conv output %s" % (tuple(dy.shape), (B, d.Ho, d.Wo, Cout)))
    d.compute = compute
    tabs = _xf_args(d, xf, B, H, W, C)
    _f32(dw, KH, KW, C, Cout)
    if db is not None:
        _f32(db, Cout)
    return (d, x, dy, tabs, dw, db)


def wgrad2_on():
    """The LDS-DMA weight-gradient kernel (csrc/conv_wgrad.hip, conv_wgrad2_kernel) takes layers whose operands are both final
    bf16 tensors; HDRSKY_WGRAD2=0 keeps everything on the register-staged kernel (A/B hook)."""
    return HOOKS.H.wgrad2


def _fill_wgrad_job(j, job):
    d, x, dy, tabs, dw, db = job[:6]
    if len(job) > 6:
        offs, j.da_ksize, j.da_C = job[6]
        j.da_offs = _p(offs)
    j.desc = d
    j.x, j.dy, j.dw, j.db = _p(x), _p(dy), _p(dw), _p(db)
    j.in_scale, j.in_shift, j.in_part, j.in_gamma, j.in_beta = [_p(t) for t in tabs]
    j.x_bf16, j.dy_bf16 = int(x.dtype == torch.bfloat16), int(dy.dtype == torch.bfloat16)


def _materialise_bf16_operand(job):
    """A weight-gradient job whose input x (fp32, or a raw conv output stored as bf16) still needs its operand transform
    (InstanceNorm / BatchNorm affine + activation), rewritten onto the final bf16 tensor x' = hdrsky_act_bf16(x): the LDS-DMA
    kernel copies its tiles without touching a register.  Only for the layers that kernel takes - the library's own answer
    (hdrsky_wgrad2_eligible); everything else is returned unchanged."""
    if len(job) > 6:
        return job
    d, x, dy, tabs, dw, db = job
    x16 = x.dtype == torch.bfloat16
    if x16 and d.in_mode == L.IN_NONE and d.in_slope == 1.0:
        return job                   # a final bf16 activation already
    j = L.WgradJob()
    _fill_wgrad_job(j, job)
    B, H, W, C = x.shape
    # (hdrsky_act_bf16 takes up to 1024 channels; the LDS-DMA kernel's own limit is wider since the 1x1 gradient on a gathered
    # operand - a transform-carrying operand beyond that stays on the register-staged kernel: ADVICE r4)
    if C > 1024 or not L.load().hdrsky_wgrad2_eligible(ctypes.byref(j), 1):
        return job
    xb = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=x.device)
    in_scale, in_shift, in_part, in_gamma, in_beta = tabs
    L.check(L.load().hdrsky_act_bf16(_p(x), int(x16), B, H * W, C, d.in_mode, _p(in_scale), _p(in_shift), d.ss_bstride, _p(in_part),
                                     d.in_nparts, _p(in_gamma), _p(in_beta), d.in_eps, d.in_slope, _p(xb), _stream()), "act_bf16")
    d2 = L.ConvDesc()
    ctypes.memmove(ctypes.byref(d2), ctypes.byref(d), ctypes.sizeof(d))
    d2.in_mode, d2.in_slope, d2.ss_bstride, d2.in_nparts = L.IN_NONE, 1.0, 0, 0
    return (d2, xb, dy, (None, None, None, None, None), dw, db)


def conv2d_wgrad_multi(jobs, deterministic=True):
    """Weight gradients of several independent conv layers in as few launches as the library can manage.
    deterministic (default): the split-K partials go through a scratch buffer and are added in a fixed order - the
    gradients are bit-reproducible; False: fp32 atomics straight into dw (no scratch, arrival-order summation)."""
    if deterministic and HOOKS.H.da_wgrad_region:
        jobs = [job for job in jobs if not (len(job) > 6 and _da_wgrad_region(job))]
    if not jobs:
        return
    if deterministic and wgrad2_on() and not HOOKS.H.wgrad_atomic:
        jobs = [_materialise_bf16_operand(job) for job in jobs]
    arr = (L.WgradJob * len(jobs))()
    for i, job in enumerate(jobs):
        _fill_wgrad_job(arr[i], job)
    lib = L.load()
    if not deterministic or HOOKS.H.wgrad_atomic:     # (HDRSKY_WGRAD_ATOMIC: an A/B switch)
        L.check(lib.hdrsky_conv2d_wgrad_multi(arr, len(jobs), _stream()), "conv2d_wgrad_multi")
        return
    nbytes = int(lib.hdrsky_conv2d_wgrad_ws_bytes(arr, len(jobs)))
    if nbytes <= 0:
        raise L.HdrSkyError("conv2d_wgrad_multi: unsupported layer geometry")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=jobs[0][1].device)
    L.check(lib.hdrsky_conv2d_wgrad_multi_det(arr, len(jobs), _p(ws), nbytes, _stream()), "conv2d_wgrad_multi_det")
    if TRACE is not None:
        flop = sum(2.0 * j[0].B * j[0].Ho * j[0].Wo * j[0].KH * j[0].KW * j[0].Cin * j[0].Cout for j in jobs)
        d0 = jobs[0][0]
        _LABEL[0] = ", ".join(WG_NAMES.get(j[4].data_ptr(), "?") for j in jobs)
        nbuf = ctypes.create_string_buffer(1024)
        lib.hdrsky_conv2d_wgrad_kernel_names(arr, len(jobs), nbuf, 1024)
        _trace("wgrad", "%s (%d layers in one call)" % (nbuf.value.decode() or "conv_wgrad_kernel + wgrad_reduce_kernel", len(jobs)),
               "%d x e.g. %dx%d %d->%d @%dx%d B=%d" % (len(jobs), d0.KH, d0.KW, d0.Cin, d0.Cout, d0.Ho, d0.Wo, d0.B), flop,
               lambda a_=arr, n_=len(jobs), w_=ws, k_=jobs: L.check(
                   lib.hdrsky_conv2d_wgrad_multi_det(a_, n_, _p(w_), w_.numel(), _stream()), "conv2d_wgrad_multi_det"))


def norm_apply(x, stats: Stats, gamma, beta, slope=1.0, residual=None, pool=False, eps=IN_EPS):
    """y = leaky(IN(x)) [+ residual]; optionally also the 2x2 max-pool of y.  Returns y or (y, ypool).  x: fp32, or the raw conv
    output as the single-product mode stores it (bfloat16; y is fp32 either way)."""
    x16 = _raw(x)
    B, H, W, C = x.shape
    _f32(stats.part, B, stats.nparts, 2, C)
    if stats.count != H * W:
        raise ValueError("partials/count mismatch")
    _f32(gamma, C); _f32(beta, C)
    if residual is not None:
        _f32(residual, B, H, W, C)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    yp = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device) if pool else None
    L.check(L.load().hdrsky_norm_apply(_p(x), x16, _p(stats.part), stats.nparts, _p(gamma), _p(beta), eps, slope,
                                       _p(residual), _p(y), _p(yp), B, H, W, C, _stream()), "norm_apply")
    return (y, yp) if pool else y




def in_xf(stats: Stats, gamma, beta, slope, eps=IN_EPS, pair=None):
    """The fused operand transform leaky(InstanceNorm(x)) for the conv / weight-gradient that consumes the raw tensor x:
    the tile partials themselves (every workgroup of the consumer derives the tables in its prologue) while a sample has
    few tiles, tables computed once by hdrsky_in_affine (same formula, equal to an ulp or two) from 64 tiles (tuning hook HDRSKY_INXF_AFFINE_MIN) per
    sample on."""
    B, nparts, _, C = stats.part.shape
    g2, b2 = pair if pair is not None else (None, None)      # a PAIRED tensor: the second half of the batch is another layer's output
    if nparts < HOOKS.H.inxf_affine_min:
        return InXf(mode=L.IN_PARTIALS, slope=slope, stats=stats, gamma=gamma, beta=beta, eps=eps, gamma2=g2, beta2=b2)
    scale = torch.empty((B, C), dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    if pair is not None:
        L.check(L.load().hdrsky_in_affine_pair(_p(stats.part), nparts, B, C, stats.count, _p(_f32(gamma, C)), _p(_f32(beta, C)), _p(_f32(g2, C)),
                                               _p(_f32(b2, C)), eps, _p(scale), _p(shift), _stream()), "in_affine_pair")
        return InXf(mode=L.IN_AFFINE, slope=slope, scale=scale, shift=shift)
    L.check(L.load().hdrsky_in_affine(_p(stats.part), nparts, B, C, stats.count, _p(_f32(gamma, C)), _p(_f32(beta, C)), eps,
                                      _p(scale), _p(shift), _stream()), "in_affine")
    return InXf(mode=L.IN_AFFINE, slope=slope, scale=scale, shift=shift)


def in_finalize(stats: Stats, gamma, beta, B, C, eps=IN_EPS):
    """(mean, rstd, scale, shift) tables [B,C]."""
    _f32(stats.part, B, stats.nparts, 2, C)
    outs = [torch.empty((B, C), dtype=torch.float32, device=gamma.device) for _ in range(4)]
    L.check(L.load().hdrsky_in_finalize(_p(stats.part), stats.nparts, B, C, stats.count, _p(_f32(gamma, C)),
                                        _p(_f32(beta, C)), eps, *[_p(o) for o in outs], _stream()), "in_finalize")
    return outs


def bn_eval_affine(gamma, beta, mm, mv, eps=IN_EPS):
    C = gamma.numel()
    for t in (gamma, beta, mm, mv):
        _f32(t, C)
    scale, shift = torch.empty_like(gamma), torch.empty_like(gamma)
    L.check(L.load().hdrsky_bn_eval_affine(_p(gamma), _p(beta), _p(mm), _p(mv), eps, C, _p(scale), _p(shift), _stream()),
            "bn_eval_affine")
    return scale, shift


def norm_act_bwd(x, stats: Stats, gamma, beta, slope, dy, pooled, eps=IN_EPS, want_sums=False, dgamma=None, dbeta=None,
                 sums=None, out_bf16=False, pair=None):
    """dx of y = leaky(IN(x)) [-> maxpool2x2]; dy is the gradient wrt y (or wrt the pooled y).
    sums [B,2,C] (given, or allocated when want_sums): per-sample (d beta, d gamma) terms - reduce them over the batch with
    DgbReducer for bit-reproducible gradients; dgamma / dbeta: accumulated by fp32 atomics instead (arrival order).
    x: fp32 or bfloat16 (the raw conv output as stored)."""
    x16 = _raw(x)
    B, H, W, C = x.shape
    _f32(stats.part, B, stats.nparts, 2, C)
    # dy: fp32, or bf16 (the output of a data-gradient conv that nothing else reads: out_bf16 of conv2d_dgrad)
    (_bf16 if dy.dtype == torch.bfloat16 else _f32)(dy, B, H // 2 if pooled else H, W // 2 if pooled else W, C)
    # out_bf16: dx only feeds a data-gradient conv / a weight gradient, which round it to bf16 anyway (HDRSKY_BF16 mode)
    dx = torch.empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    if sums is not None:
        _f32(sums, B, 2, C)
        want_sums = False
    elif want_sums:
        sums = torch.empty((B, 2, C), dtype=torch.float32, device=x.device)
    S = L.load().hdrsky_norm_act_bwd_nslices(B // 2 if pair is not None else B, H, W, C, int(pooled))      # (a paired tensor: sliced like one layer's launch)
    ws = torch.empty((B, S, 2, C), dtype=torch.float32, device=x.device) if S > 1 else None
    if pair is not None:      # a PAIRED tensor: (gamma, beta) for the first half of the batch, pair = (gamma2, beta2) for the second
        if dgamma is not None or dbeta is not None:
            raise ValueError("norm_act_bwd pair: per-sample sums only")
        L.check(L.load().hdrsky_norm_act_bwd_pair(_p(x), _p(stats.part), stats.nparts, _p(_f32(gamma, C)), _p(_f32(beta, C)), _p(_f32(pair[0], C)),
                                                  _p(_f32(pair[1], C)), eps, slope, _p(dy), int(pooled), _p(dx),
                                                  int(out_bf16) | (2 if dy.dtype == torch.bfloat16 else 0) | (4 if x16 else 0), _p(sums), _p(ws),
                                                  B, H, W, C, _stream()), "norm_act_bwd_pair")
        return (dx, sums) if want_sums else dx
    L.check(L.load().hdrsky_norm_act_bwd(_p(x), _p(stats.part), stats.nparts, _p(_f32(gamma, C)), _p(_f32(beta, C)),
                                         eps, slope, _p(dy), int(pooled), _p(dx), int(out_bf16) | (2 if dy.dtype == torch.bfloat16 else 0) | (4 if x16 else 0), _p(sums), _p(dgamma),
                                         _p(dbeta), _p(ws), B, H, W, C, _stream()),
            "norm_act_bwd")
    return (dx, sums) if want_sums else dx


class PackedFC:
    """bf16 images of a Dense kernel [K,N]: packed [K/8][N][8] (forward) and natural [K][N] (dgrad)."""

    def __init__(self, w, precise=True, need_dgrad=True):
        _f32(w)
        self.K, self.N = w.shape
        dev = w.device
        mk = lambda: torch.empty(self.K * self.N, dtype=torch.bfloat16, device=dev)
        self.pk_hi = mk()
        self.pk_lo = mk() if precise else None
        self.nat_hi = mk() if need_dgrad else None
        self.nat_lo = mk() if (need_dgrad and precise) else None
        self.repack(w)

    def repack(self, w):
        L.check(L.load().hdrsky_fc_pack_weights(_p(w), self.K, self.N, _p(self.pk_hi), _p(self.pk_lo), _p(self.nat_hi),
                                                _p(self.nat_lo), _stream()), "fc_pack_weights")


FC_MAX_ROWS = 32   # rows one fc launch takes (the reference's batch size); larger batches go in slices of 32 rows


def fc_fwd(x, pf: PackedFC, compute=BF16):
    """Split-R partial products [nsplit, M, N] of x[M,K] @ W[K,N]."""
    M, K = x.shape
    _f32(x, M, pf.K)
    if M > FC_MAX_ROWS:
        return torch.cat([fc_fwd(x[i:i + FC_MAX_ROWS], pf, compute) for i in range(0, M, FC_MAX_ROWS)], dim=1)
    lib = L.load()
    ns = lib.hdrsky_fc_nsplit(K)
    out = torch.empty((ns, M, pf.N), dtype=torch.float32, device=x.device)
    L.check(lib.hdrsky_fc_fwd(_p(x), _p(pf.pk_hi), _p(pf.pk_lo), M, K, pf.N, ns, compute, _p(out), _stream()), "fc_fwd")
    return out


def fc_dgrad(dy, pf: PackedFC, compute=BF16):
    """Split-R partial products [nsplit, M, K] of dy[M,N] @ W[K,N]^T."""
    M, N = dy.shape
    _f32(dy, M, pf.N)
    if M > FC_MAX_ROWS:
        return torch.cat([fc_dgrad(dy[i:i + FC_MAX_ROWS], pf, compute) for i in range(0, M, FC_MAX_ROWS)], dim=1)
    lib = L.load()
    ns = lib.hdrsky_fc_nsplit(N)
    out = torch.empty((ns, M, pf.K), dtype=torch.float32, device=dy.device)
    L.check(lib.hdrsky_fc_dgrad(_p(dy), _p(pf.nat_hi), _p(pf.nat_lo), M, pf.K, N, ns, compute, _p(out), _stream()),
            "fc_dgrad")
    return out


def _fc_fin(x, pf, compute, dgrad, bias, relu, mask_src, zero_word):
    """fc_fwd / fc_dgrad + fc_finalize: ONE launch (hdrsky_fc_fwd_fin / hdrsky_fc_dgrad_fin - the last workgroup of a column block
    adds the reduction slices in slice order: bit-identical) where the rows fit a launch and HDRSKY_FC_FIN is not 0, else the two."""
    M = x.shape[0]
    if M > FC_MAX_ROWS or not HOOKS.H.fc_fin:
        return fc_finalize((fc_dgrad if dgrad else fc_fwd)(x, pf, compute), bias, relu, mask_src, zero_word)
    R, O = (pf.N, pf.K) if dgrad else (pf.K, pf.N)
    _f32(x, M, R)
    lib = L.load()
    ns = lib.hdrsky_fc_nsplit(R)
    part = torch.empty((ns, M, O), dtype=torch.float32, device=x.device)
    y = torch.empty((M, O), dtype=torch.float32, device=x.device)
    if bias is not None:
        _f32(bias, O)
    if mask_src is not None:
        _f32(mask_src, M, O)
    # the tickets of this call site: one set per (layer, direction, stream) - launches on one stream are ordered, and a captured
    # graph replays the launch with the same words
    cnt = pf.__dict__.setdefault("_tickets", {})
    key = (bool(dgrad), torch.cuda.current_stream().cuda_stream)
    if key not in cnt:
        cnt[key] = torch.zeros(((O + 63) // 64,), dtype=torch.int32, device=x.device)
    if dgrad:
        L.check(lib.hdrsky_fc_dgrad_fin(_p(x), _p(pf.nat_hi), _p(pf.nat_lo), M, pf.K, pf.N, ns, compute, _p(part), _p(cnt[key]), _p(bias),
                                        int(relu), _p(mask_src), _p(y), _p(zero_word), _stream()), "fc_dgrad_fin")
    else:
        L.check(lib.hdrsky_fc_fwd_fin(_p(x), _p(pf.pk_hi), _p(pf.pk_lo), M, pf.K, pf.N, ns, compute, _p(part), _p(cnt[key]), _p(bias),
                                      int(relu), _p(mask_src), _p(y), _p(zero_word), _stream()), "fc_fwd_fin")
    return y


def fc_fwd_fin(x, pf: PackedFC, compute=BF16, bias=None, relu=False, mask_src=None, zero_word=None):
    """act(bias + x @ W) [M, N] (Keras Dense, sunpose_net.py:48-51): fc_finalize(fc_fwd(...)) in one launch."""
    return _fc_fin(x, pf, compute, False, bias, relu, mask_src, zero_word)


def fc_dgrad_fin(dy, pf: PackedFC, compute=BF16, mask_src=None):
    """mask(dy @ W^T) [M, K] (tf.gradients through Dense, grad_cam.py:31): fc_finalize(fc_dgrad(...)) in one launch."""
    return _fc_fin(dy, pf, compute, True, None, False, mask_src, None)


def fc_finalize(part, bias=None, relu=False, mask_src=None, zero_word=None):
    """zero_word: an int32[1] tensor this launch clears (softmax_head's max accumulator further down the chain)."""
    ns, M, N = part.shape
    _f32(part)
    y = torch.empty((M, N), dtype=torch.float32, device=part.device)
    if bias is not None:
        _f32(bias, N)
    if mask_src is not None:
        _f32(mask_src, M, N)
    L.check(L.load().hdrsky_fc_finalize(_p(part), ns, M, N, _p(bias), int(relu), _p(mask_src), _p(y), _p(zero_word),
                                        _stream()), "fc_finalize")
    return y


def global_max(x, gmax_bits=None):
    """int32[1] holding the bit pattern of max(x) for a non-negative fp32 tensor (tf.reduce_max, generator.py:160): what
    softmax_head leaves in its max accumulator, for a sun-position map that is an input of the step."""
    _f32(x)
    if gmax_bits is None:
        gmax_bits = torch.empty(1, dtype=torch.int32, device=x.device)
    zero_(gmax_bits)
    L.check(L.load().hdrsky_global_max(_p(x), x.numel(), _p(gmax_bits), _stream()), "global_max")
    return gmax_bits


def softmax_head(part, bias, gmax_bits=None):
    """z = relu(sum part + bias), cmf = softmax(z); gmax_bits (int32[1], zeroed by the caller) tracks max(cmf)."""
    ns, M, N = part.shape
    _f32(part); _f32(bias, N)
    z = torch.empty((M, N), dtype=torch.float32, device=part.device)
    cmf = torch.empty_like(z)
    L.check(L.load().hdrsky_softmax_head(_p(part), ns, M, N, _p(bias), _p(z), _p(cmf), _p(gmax_bits), _stream()),
            "softmax_head")
    return z, cmf


def softmax_head_pick(part, bias, gmax_bits=None, pick_src=None):
    """softmax_head + softmax_pick_bwd in one launch -> (z, cmf, dz); pick_src None: the row's own cmf picks the class."""
    ns, M, N = part.shape
    _f32(part); _f32(bias, N)
    if pick_src is not None:
        _f32(pick_src, M, N)
    z = torch.empty((M, N), dtype=torch.float32, device=part.device)
    cmf, dz = torch.empty_like(z), torch.empty_like(z)
    L.check(L.load().hdrsky_softmax_head_pick(_p(part), ns, M, N, _p(bias), _p(z), _p(cmf), _p(gmax_bits), _p(pick_src), _p(dz),
                                              None, _stream()), "softmax_head_pick")
    return z, cmf, dz


def softmax_pick_bwd(cmf, z, pick_src):
    M, N = cmf.shape
    _f32(cmf); _f32(z, M, N); _f32(pick_src, M, N)
    dz = torch.empty_like(cmf)
    idx = torch.empty((M,), dtype=torch.int32, device=cmf.device)
    L.check(L.load().hdrsky_softmax_pick_bwd(_p(cmf), _p(z), _p(pick_src), M, N, _p(dz), _p(idx), _stream()),
            "softmax_pick_bwd")
    return dz, idx


def spatial_sum(x, scale=1.0):
    B, H, W, C = x.shape
    _f32(x)
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    L.check(L.load().hdrsky_spatial_sum(_p(x), B, H * W, C, scale, _p(out), _stream()), "spatial_sum")
    return out


def _cam_weights(A, w):
    B, H, W, C = A.shape
    _f32(A)
    if isinstance(w, Stats):
        _f32(w.part, B, w.nparts, 2, C)
        return w.part, w.nparts
    if w.dim() == 4:        # the activation gradient itself [B,h,w,C] (a small map): its spatial sum is taken in the launch
        if w.shape[0] != B or w.shape[3] != C or w.shape[1] * w.shape[2] > 256:
            raise ValueError("grad_cam_map: gradient map %s" % (tuple(w.shape),))
        return _f32(w), -(w.shape[1] * w.shape[2])
    return _f32(w, B, C), 0


def grad_cam_map(A, w, scale=1.0):
    """cam = relu(sum_c w_c * A[..., c]).  w: [B,C] table, or the Stats of the conv that produced the activation
    gradient (its per-tile sums are the GAP numerator)."""
    B, H, W, C = A.shape
    wp, nparts = _cam_weights(A, w)
    cam = torch.empty((B, H, W, 1), dtype=torch.float32, device=A.device)
    L.check(L.load().hdrsky_grad_cam(_p(A), _p(wp), nparts, scale, B, H * W, C, _p(cam), _stream()), "grad_cam")
    return cam


def grad_cam_maps(jobs):
    """The three maps of one Grad-CAM sweep, [(A, w, scale)] x 3 as for grad_cam_map, in one launch (same values)."""
    if len(jobs) != 3:
        raise ValueError("grad_cam_maps: three (A, w, scale) jobs")
    B = jobs[0][0].shape[0]
    wps, cams = [], []
    for A, w, _ in jobs:
        if A.shape[0] != B:
            raise ValueError("grad_cam_maps: batch sizes differ")
        wps.append(_cam_weights(A, w))
        cams.append(torch.empty(tuple(A.shape[:3]) + (1,), dtype=torch.float32, device=A.device))
    vp, ci, cf = ctypes.c_void_p * 3, ctypes.c_int * 3, ctypes.c_float * 3
    L.check(L.load().hdrsky_grad_cam3(vp(*[_p(j[0]) for j in jobs]), vp(*[_p(wp[0]) for wp in wps]), ci(*[wp[1] for wp in wps]),
                                      cf(*[float(j[2]) for j in jobs]), ci(*[j[0].shape[1] * j[0].shape[2] for j in jobs]),
                                      ci(*[j[0].shape[3] for j in jobs]), vp(*[_p(c) for c in cams]), B, _stream()),
            "grad_cam3")
    return tuple(cams)


def plz_build(ldr, cam1, cam2, cam3):
    B, H, W, _ = ldr.shape
    _f32(ldr, B, H, W, 3); _f32(cam1, B, H, W, 1); _f32(cam2, B, H // 2, W // 2, 1); _f32(cam3, B, H // 4, W // 4, 1)
    plz = torch.empty((B, H, W, 6), dtype=torch.float32, device=ldr.device)
    L.check(L.load().hdrsky_plz_build(_p(ldr), _p(cam1), _p(cam2), _p(cam3), B, H, W, _p(plz), _stream()), "plz_build")
    return plz


HEAD_SLICES = 16


def dense_heads(x, scale, shift, slope, kg, kb):
    """Stage 1 of the sunRadNet gamma/beta heads: per-slice partial dot products [B, HEAD_SLICES, 2]."""
    B = x.shape[0]
    C = x.shape[-1]
    F = x[0].numel()
    _f32(x); _f32(kg, F, 1); _f32(kb, F, 1)
    if scale is not None:
        _f32(scale, C); _f32(shift, C)
    part = torch.empty((B, HEAD_SLICES, 2), dtype=torch.float32, device=x.device)
    L.check(L.load().hdrsky_dense_heads(_p(x), _p(scale), _p(shift), slope, B, F, C, _p(kg), _p(kb), HEAD_SLICES,
                                        _p(part), _stream()), "dense_heads")
    return part


def sun_rad(cmf, gmax_bits, head_part, bg, bb, H, W):
    """gamma/beta = sigmoid(sum head_part + bias); Dirac-delta radiance (linear, x3) and its log-compressed image."""
    B, P = cmf.shape
    _f32(cmf, B, H * W); _f32(head_part, B, head_part.shape[1], 2); _f32(bg, 1); _f32(bb, 1)
    lin = torch.empty((B, H, W, 3), dtype=torch.float32, device=cmf.device)
    gam = torch.empty_like(lin)
    g = torch.empty((B, 1, 1, 1), dtype=torch.float32, device=cmf.device)
    b = torch.empty_like(g)
    L.check(L.load().hdrsky_sun_rad(_p(cmf), _p(gmax_bits), _p(head_part), head_part.shape[1], _p(bg), _p(bb), B, P,
                                    _p(g), _p(b), _p(lin), _p(gam), _stream()), "sun_rad")
    return lin, gam, g, b


def blend(sky_gamma, sun_gamma, thr=0.12, extras=True):
    _f32(sky_gamma); _f32(sun_gamma, *sky_gamma.shape)
    if sky_gamma.shape[-1] != 3:
        raise ValueError("blend expects 3 channels")
    npix = sky_gamma.numel() // 3
    yg, yl = torch.empty_like(sky_gamma), torch.empty_like(sky_gamma)
    al = sl = ul = None
    if extras:
        al, sl, ul = torch.empty_like(sky_gamma), torch.empty_like(sky_gamma), torch.empty_like(sky_gamma)
    L.check(L.load().hdrsky_blend(_p(sky_gamma), _p(sun_gamma), npix, thr, _p(yg), _p(yl), _p(al), _p(sl), _p(ul),
                                  _stream()), "blend")
    return yg, yl, al, sl, ul


def tonemap(x, decompress):
    _f32(x)
    y = torch.empty_like(x)
    L.check(L.load().hdrsky_tonemap(_p(x), _p(y), x.numel(), int(decompress), _stream()), "tonemap")
    return y


# ------------------------------------------------------------------------------------------------
# training-step front ends (csrc/train_ops.hip, csrc/conv_wgrad.hip)
# ------------------------------------------------------------------------------------------------
def leaky_relu(x, slope=0.0):
    """ops.relu (ops.py:324-329) for slope 0; LeakyReLU otherwise."""
    x = _f32(x)
    y = torch.empty_like(x)
    L.check(L.load().hdrsky_leaky_relu(_p(x), _p(y), x.numel(), float(slope), _stream()), "leaky_relu")
    return y


def conv_dgrad_desc(fwd):
    d = L.ConvDesc()
    L.check(L.load().hdrsky_conv_desc_init_dgrad(d, fwd), "conv_desc_init_dgrad")
    return d


def conv2d_dgrad(dy, pwT: PackedConv, fwd_desc, residual=None, compute=BF16, want_stats=False, **kw):
    """Gradient wrt the conv-input domain of the forward conv `fwd_desc` (pwT = PackedConv(w, transpose_flip=True)).
    kw: out_bf16 / mask_bf16 / mask_slope of conv2d (the activation backward in front of the NEXT data gradient)."""
    return conv2d(dy, pwT, None, desc=conv_dgrad_desc(fwd_desc), residual=residual, compute=compute, want_stats=want_stats,
                  **kw)


def _empty_like_shape(t, shape):
    return torch.empty(shape, dtype=torch.float32, device=t.device)


def bn_train_finalize(stats: Stats, gamma, beta, B, C, moving_mean=None, moving_var=None, eps=IN_EPS, momentum=0.99,
                      scale_rows=None, shift_rows=None):
    """Batch statistics over (N,H,W) from conv partials -> (mean, rstd, scale, shift) [C]; updates moving stats in place.
    scale_rows / shift_rows [R,C] (given together): scale / shift are written as R identical rows into them instead (the
    per-sample affine tables of a batch that holds several BatchNorm groups)."""
    _f32(stats.part, B, stats.nparts, 2, C)
    outs = [torch.empty((C,), dtype=torch.float32, device=gamma.device) for _ in range(2)]
    rows = 1
    if scale_rows is not None:
        rows = scale_rows.shape[0]
        _f32(scale_rows, rows, C); _f32(shift_rows, rows, C)
        outs += [scale_rows, shift_rows]
    else:
        outs += [torch.empty((C,), dtype=torch.float32, device=gamma.device) for _ in range(2)]
    L.check(L.load().hdrsky_bn_train_finalize(_p(stats.part), B * stats.nparts, C, B * stats.count, _p(_f32(gamma, C)),
                                              _p(_f32(beta, C)), eps, momentum, _p(moving_mean), _p(moving_var),
                                              *[_p(o) for o in outs], rows, _stream()), "bn_train_finalize")
    return outs


def zero_(t):
    """Zero-fill of a contiguous tensor on the current stream (a kernel: hipGraph memset nodes only worked in the first
    replay of a captured segment, see hdrsky_zero)."""
    if not (torch.is_tensor(t) and t.is_cuda and t.is_contiguous()):
        raise ValueError("expected a contiguous CUDA tensor")
    L.check(L.load().hdrsky_zero(_p(t), t.numel() * t.element_size(), _stream()), "zero")
    return t


def bn_act_bwd(x, dy, mean, rstd, gamma, beta, slope, dgamma=None, dbeta=None, out=None, out_bf16=False, sync=None):
    """Gradient through LeakyReLU(BatchNorm(x)) in training mode.  sync (parallel.BatchSync): the batch statistics ran over
    every replica's batch - the two means of the formula then run over the all-gathered partial blocks, d gamma / d beta
    over this replica's (the gradient exchange sums them).  x: fp32 or bfloat16 (the raw conv output as stored)."""
    x16 = _raw(x)
    if x16 and sync is not None:
        raise ValueError("bn_act_bwd: bf16 storage of the raw conv output is not supported with batch statistics over replicas")
    dy16 = dy.dtype == torch.bfloat16       # a data-gradient conv's bf16 output (single-replica path only)
    if dy16 and sync is not None:
        raise ValueError("bn_act_bwd: a bf16 incoming gradient is not supported with batch statistics over replicas")
    (_bf16 if dy16 else _f32)(dy, *x.shape)
    C = x.shape[-1]
    npix = x.numel() // C
    lib = L.load()
    nb = lib.hdrsky_bn_bwd_nblocks()
    ws = torch.empty((2 * nb * C + 2 * C,), dtype=torch.float32, device=x.device)
    if out_bf16:
        dx = _bf16(out, *x.shape) if out is not None else torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    else:
        dx = _f32(out, *x.shape) if out is not None else torch.empty(x.shape, dtype=torch.float32, device=x.device)
    args = (_p(x), _p(dy), _p(_f32(mean, C)), _p(_f32(rstd, C)), _p(_f32(gamma, C)), _p(_f32(beta, C)), slope, npix, C)
    if sync is None:
        L.check(lib.hdrsky_bn_act_bwd(*args, _p(ws), _p(dgamma), _p(dbeta), _p(dx), int(out_bf16) | (2 if dy16 else 0) | (4 if x16 else 0), _stream()),
                "bn_act_bwd")
        return dx
    part = ws[:2 * nb * C].view(nb, 2, C)
    L.check(lib.hdrsky_bn_act_bwd_reduce(*args, _p(part), _stream()), "bn_act_bwd_reduce")
    allp = sync.gather_rows(part)                       # [world * nb, 2, C], replica order
    L.check(lib.hdrsky_bn_act_bwd_apply(*args, _p(allp), allp.shape[0], float(npix) * sync.world, _p(part), nb,
                                        _p(ws[2 * nb * C:]), _p(dgamma), _p(dbeta), _p(dx), int(out_bf16), _stream()),
            "bn_act_bwd_apply")
    return dx


def affine_act_bwd(x, dy, scale, shift, slope, out_bf16=False):
    odt = torch.bfloat16 if out_bf16 else torch.float32
    dy16 = dy.dtype == torch.bfloat16       # a data-gradient conv's bf16 output
    if x.dtype == torch.bfloat16 and scale is None and shift is None:      # plain activation backward on an ACTIVATED bf16 tensor
        _bf16(x); (_bf16 if dy16 else _f32)(dy, *x.shape)
        dx = torch.empty(dy.shape, dtype=odt, device=dy.device)
        L.check(L.load().hdrsky_act_bwd_bf16(_p(x), _p(dy), slope, x.numel(), _p(dx), int(out_bf16) | (2 if dy16 else 0), _stream()),
                "act_bwd_bf16")
        return dx
    x16 = _raw(x); (_bf16 if dy16 else _f32)(dy, *x.shape)          # (x bf16 + an affine: a raw conv output as stored)
    C = x.shape[-1]
    if x16 and (C & 3):
        raise ValueError("affine_act_bwd: bf16 storage wants C % 4 == 0")
    dx = torch.empty(x.shape, dtype=odt, device=x.device)
    L.check(L.load().hdrsky_affine_act_bwd(_p(x), _p(dy), _p(scale), _p(shift), slope, x.numel(), C, _p(dx),
                                           int(out_bf16) | (2 if dy16 else 0) | (4 if x16 else 0), _stream()), "affine_act_bwd")
    return dx


def maxpool(y, want_bf16=False):
    """2x2 max-pool.  A bfloat16 map returns (fp32 pool, bf16 pool or None)."""
    B, H, W, C = y.shape
    p = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=y.device)
    if y.dtype == torch.bfloat16:
        _bf16(y)
        pb = torch.empty((B, H // 2, W // 2, C), dtype=torch.bfloat16, device=y.device) if want_bf16 else None
        L.check(L.load().hdrsky_maxpool_fwd_bf16(_p(y), B, H, W, C, _p(p), _p(pb), _stream()), "maxpool_fwd_bf16")
        return p, pb
    _f32(y)
    L.check(L.load().hdrsky_maxpool_fwd(_p(y), B, H, W, C, _p(p), _stream()), "maxpool_fwd")
    return p


def maxpool_relu_bwd(y, dp, out_bf16=False):
    B, H, W, C = y.shape
    _f32(dp, B, H // 2, W // 2, C)
    dy = torch.empty((B, H, W, C), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=y.device)
    if y.dtype == torch.bfloat16:
        L.check(L.load().hdrsky_maxpool_relu_bwd_bf16(_p(_bf16(y)), _p(dp), B, H, W, C, _p(dy), int(out_bf16), _stream()),
                "maxpool_relu_bwd_bf16")
        return dy
    if out_bf16:
        raise ValueError("bf16 output: bf16 activation input only")
    _f32(y)
    L.check(L.load().hdrsky_maxpool_relu_bwd(_p(y), _p(dp), B, H, W, C, _p(dy), _stream()), "maxpool_relu_bwd")
    return dy


def maxpool_relu_l1_bwd(y, pool, target, dp, wl, wg, loss_slot, out_bf16=True):
    """maxpool_relu_bwd(y, dp') with dp' = (dp or 0) + the gradient of wl * mean|pool - target| (weight wg), the term's value added
    to loss_slot: the L1 launch of a VGG16 block's pooled features folded into its pool backward.  y: bfloat16 activation."""
    B, H, W, C = y.shape
    _bf16(y); _f32(pool, B, H // 2, W // 2, C); _f32(target, B, H // 2, W // 2, C)
    if dp is not None:
        _f32(dp, B, H // 2, W // 2, C)
    dy = torch.empty((B, H, W, C), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=y.device)
    L.check(L.load().hdrsky_maxpool_relu_l1_bwd_bf16(_p(y), _p(pool), _p(target), _p(dp), B, H, W, C, wl, wg, _p(loss_slot), _p(dy),
                                                     int(out_bf16), _stream()), "maxpool_relu_l1_bwd_bf16")
    return dy


def up2x(a, b=None):
    B, H, W, C = a.shape
    _f32(a)
    if b is not None:
        _f32(b, *a.shape)
    y = torch.empty((B, 2 * H, 2 * W, C), dtype=torch.float32, device=a.device)
    L.check(L.load().hdrsky_up2x_fwd(_p(a), _p(b), B, H, W, C, _p(y), _stream()), "up2x_fwd")
    return y


def deconv_materialised(compute):
    """Whether the training step runs its resize-deconvolutions as (operand written once as bf16) + plain conv + plain
    weight gradient - single-product mode; HDRSKY_DECONV_MAT=0 keeps the resize fused into the staging of both (A/B hook:
    step -0.4 %, all of it from the weight gradients; the forward pass alone gains nothing, engine.decode stays fused)."""
    return compute == BF16 and HOOKS.H.deconv_mat


def up2x_act_bf16(x, xf: Optional[InXf] = None):
    """bf16 [B,2H,2W,C] = resize2x(leaky(IN(x))) - the materialised operand of a resize-deconvolution (xf: IN_PARTIALS
    transform of the producing conv, or None for an activation); feed it to conv2d / wgrad_job with upsample=1.
    x: fp32 or bfloat16 (the raw conv output as stored)."""
    x16 = _raw(x)
    B, H, W, C = x.shape
    y = torch.empty((B, 2 * H, 2 * W, C), dtype=torch.bfloat16, device=x.device)
    part = gamma = beta = None
    nparts, eps, slope = 0, IN_EPS, 1.0
    if xf is not None:
        if xf.mode != L.IN_PARTIALS:
            raise ValueError("up2x_act_bf16: InstanceNorm-partials transform (or none) only")
        st = xf.stats
        part, nparts, eps, slope = _f32(st.part, B, st.nparts, 2, C), st.nparts, float(xf.eps), float(xf.slope)
        gamma, beta = _f32(xf.gamma, C), _f32(xf.beta, C)
        if st.count != H * W:
            raise ValueError("partials were not accumulated over this tensor's H*W")
    if xf is not None and xf.gamma2 is not None:      # a PAIRED tensor
        L.check(L.load().hdrsky_up2x_xf_bf16_pair(_p(x), x16, B, H, W, C, _p(part), nparts, _p(gamma), _p(beta), _p(_f32(xf.gamma2, C)),
                                                  _p(_f32(xf.beta2, C)), eps, slope, _p(y), _stream()), "up2x_xf_bf16_pair")
        return y
    L.check(L.load().hdrsky_up2x_xf_bf16(_p(x), x16, B, H, W, C, _p(part), nparts, _p(gamma), _p(beta), eps, slope, _p(y), _stream()),
            "up2x_xf_bf16")
    return y


def up2x_bwd(dy, scale=1.0, out=None, pair_sum=False):
    """Adjoint of the bilinear 2x resize; out: ACCUMULATED into.  pair_sum: dy holds 2 B samples (a paired gradient) and
    dx[b] = adjoint(dy[b]) + adjoint(dy[b + B]) - two layers' gradients with respect to an input they share, in the order of two
    accumulating calls."""
    B, H2, W2, C = dy.shape
    (_bf16 if dy.dtype == torch.bfloat16 else _f32)(dy)      # bf16: a data-gradient conv's out_bf16 output
    if pair_sum:
        if B % 2:
            raise ValueError("up2x_bwd pair_sum: an even batch")
        B //= 2
    acc = out is not None
    dx = out if acc else torch.empty((B, H2 // 2, W2 // 2, C), dtype=torch.float32, device=dy.device)
    _f32(dx, B, H2 // 2, W2 // 2, C)
    L.check(L.load().hdrsky_up2x_bwd(_p(dy), B, H2 // 2, W2 // 2, C, scale,
                                     int(acc) | (2 if dy.dtype == torch.bfloat16 else 0) | (4 if pair_sum else 0), _p(dx), _stream()), "up2x_bwd")
    return dx


def blur3(x, sigma, transpose=False):
    B, H, W, C = x.shape
    _f32(x)
    y = torch.empty_like(x)
    L.check(L.load().hdrsky_blur3(_p(x), B, H, W, C, sigma, int(transpose), _p(y), _stream()), "blur3")
    return y


DOG_SIGMA_BASE = 1.2489996


def dog_loss(y_lin, hdr_t, weight, loss_slot, dy_out):
    """loss_slot += weight * DoG-L1(y_lin, hdr_t) (tf_utils.py:61-73, train.py:316-322); dy_out += its gradient wrt y_lin."""
    lib = L.load()
    B, H, W, C = y_lin.shape
    _f32(y_lin); _f32(hdr_t, B, H, W, C); _f32(dy_out, B, H, W, C)
    if HOOKS.H.dog_fused:     # (HDRSKY_DOG_FUSED=0: the seven staged launches)
        rc = lib.hdrsky_dog_loss(_p(y_lin), _p(hdr_t), B, H, W, C, weight, _p(loss_slot), _p(dy_out), _stream())
        if rc != L.HDRSKY_EUNSUPPORTED:
            L.check(rc, "dog_loss")
            return
    # rows too long for the one-launch kernel's LDS bands (e.g. 128x512 maps): the staged path
    up = up2x(y_lin, hdr_t)
    base = blur3(up, DOG_SIGMA_BASE)
    h = torch.empty((5,) + tuple(base.shape), dtype=torch.float32, device=base.device)
    L.check(lib.hdrsky_dog_mid(_p(base), B, 2 * H, 2 * W, C, weight, _p(h), _p(loss_slot), _stream()), "dog_mid")
    dbase = torch.empty_like(base)
    L.check(lib.hdrsky_dog_mid_bwd(_p(h), B, 2 * H, 2 * W, C, _p(dbase), _stream()), "dog_mid_bwd")
    dup = blur3(dbase, DOG_SIGMA_BASE, transpose=True)
    up2x_bwd(dup, 1.0, out=dy_out)


def l1(a, b, wl, wg, loss_slot, da=None, accumulate=False):
    _f32(a)
    if b is not None:
        _f32(b, *a.shape)
    L.check(L.load().hdrsky_l1(_p(a), _p(b), a.numel(), wl, wg, _p(loss_slot), _p(da), int(accumulate), _stream()), "l1")


def mse(x, target, wl, wg, loss_slot, want_grad=True, out=None):
    _f32(x)
    dx = (_f32(out, *x.shape) if out is not None else torch.empty_like(x)) if want_grad else None
    L.check(L.load().hdrsky_mse(_p(x), target, x.numel(), wl, wg, _p(loss_slot), _p(dx), _stream()), "mse")
    return dx


def kl(gt, cmf, loss_slot):
    B, N = cmf.shape
    _f32(gt, B, N); _f32(cmf)
    d = torch.empty_like(cmf)
    L.check(L.load().hdrsky_kl(_p(gt), _p(cmf), B, N, _p(loss_slot), _p(d), _stream()), "kl")
    return d


def softmax_bwd(cmf, dcmf, z):
    M, N = cmf.shape
    dz = torch.empty_like(cmf)
    L.check(L.load().hdrsky_softmax_bwd(_p(_f32(cmf)), _p(_f32(dcmf, M, N)), _p(_f32(z, M, N)), M, N, _p(dz), _stream()),
            "softmax_bwd")
    return dz


def blend_bwd(y_gamma, alpha, dyg, dyl):
    ds, du = torch.empty_like(y_gamma), torch.empty_like(y_gamma)
    L.check(L.load().hdrsky_blend_bwd(_p(_f32(y_gamma)), _p(_f32(alpha, *y_gamma.shape)), _p(dyg), _p(dyl), y_gamma.numel(),
                                      _p(ds), _p(du), _stream()), "blend_bwd")
    return ds, du


def head_bwd(y_gamma, alpha, dyg, dyl, din6, y_f, res_f, y_u, res_u, out=None):
    """blend_bwd + both decoder_tail_bwd in one launch; din6 [B,H,W,6] (or None): the adversarial term's gradient wrt the
    discriminator input, whose channels 3..5 are added to dyl.  Returns (dc_f, dc_u, dres_u)."""
    shp = y_gamma.shape
    for t in (alpha, y_f, res_f, y_u, res_u):
        _f32(t, *shp)
    if din6 is not None:
        _f32(din6, *(tuple(shp[:-1]) + (6,)))
    if out is not None:      # (dc_f, dc_u, dres_u) given - e.g. the two halves of one paired tensor
        dc_f, dc_u, dres_u = (_f32(t, *shp) for t in out)
    else:
        dc_f, dc_u, dres_u = torch.empty_like(y_gamma), torch.empty_like(y_gamma), torch.empty_like(y_gamma)
    L.check(L.load().hdrsky_head_bwd(_p(_f32(y_gamma)), _p(alpha), _p(dyg), _p(dyl), _p(din6), _p(y_f), _p(res_f), _p(y_u), _p(res_u),
                                     y_gamma.numel(), _p(dc_f), _p(dc_u), _p(dres_u), _stream()), "head_bwd")
    return dc_f, dc_u, dres_u


def decoder_tail_bwd(y, res, dy, want_dres):
    dc = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    L.check(L.load().hdrsky_decoder_tail_bwd(_p(_f32(y)), _p(_f32(res, *y.shape)), _p(_f32(dy, *y.shape)), y.numel(), _p(dc),
                                             _p(dres), _stream()), "decoder_tail_bwd")
    return dc, dres


def sun_rad_bwd(cmf, gmax_bits, gamma, beta, drg3, dcmf, sync=None):
    """sync (parallel.BatchSync): gmax is the maximum over every replica's batch - the maximum's gradient term and the
    tie count are then summed over the all-gathered per-replica records."""
    B, P = cmf.shape
    lib = L.load()
    scratch = torch.empty((B * P + B + 4 + 3 * B * lib.hdrsky_sun_rad_bwd_slices(P),), dtype=torch.float32, device=cmf.device)
    dpre = torch.empty((B, 2), dtype=torch.float32, device=cmf.device)
    if sync is None:
        L.check(lib.hdrsky_sun_rad_bwd(_p(cmf), _p(gmax_bits), _p(gamma), _p(beta), _p(_f32(drg3)), B, P, _p(scratch), _p(dpre),
                                       _p(_f32(dcmf, B, P)), _stream()), "sun_rad_bwd")
        return dpre
    L.check(lib.hdrsky_sun_rad_bwd_reduce(_p(cmf), _p(gmax_bits), _p(gamma), _p(beta), _p(_f32(drg3)), B, P, _p(scratch), _p(dpre),
                                          _stream()), "sun_rad_bwd_reduce")
    rec = sync.gather_rows(scratch[B * P:B * P + B + 1].view(1, B + 1))      # [world, B + 1]
    L.check(lib.hdrsky_sun_rad_bwd_apply(_p(cmf), _p(gmax_bits), _p(scratch), _p(rec), rec.shape[0], B, P, _p(_f32(dcmf, B, P)),
                                         _stream()), "sun_rad_bwd_apply")
    return dpre


def dense_heads_bwd(x, scale, shift, slope, kg, kb, dpre, dkg, dkb, dbg, dbb):
    B = x.shape[0]; C = x.shape[-1]; F = x[0].numel()
    dact = torch.empty_like(x)
    L.check(L.load().hdrsky_dense_heads_bwd(_p(_f32(x)), _p(scale), _p(shift), slope, B, F, C, _p(kg), _p(kb), _p(dpre), _p(dact),
                                            _p(dkg), _p(dkb), _p(dbg), _p(dbb), _stream()), "dense_heads_bwd")
    return dact


def slice_channels(x, c_off, c_take, scale=1.0, out=None):
    C = x.shape[-1]
    npix = x.numel() // C
    acc = out is not None
    o = out if acc else torch.empty(tuple(x.shape[:-1]) + (c_take,), dtype=torch.float32, device=x.device)
    L.check(L.load().hdrsky_slice_channels(_p(_f32(x)), npix, C, c_off, c_take, scale, int(acc), _p(o), _stream()), "slice_channels")
    return o


def pad_channels(x, Cpad, out=None):
    """[..., C] -> [..., Cpad] with zero channels appended (out: a preallocated destination)."""
    C = x.shape[-1]
    shape = tuple(x.shape[:-1]) + (Cpad,)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    _f32(out, *shape)
    L.check(L.load().hdrsky_pad_channels(_p(_f32(x)), x.numel() // C, C, Cpad, _p(out), _stream()), "pad_channels")
    return out


def concat2(a, b, out=None):
    Ca, Cb = a.shape[-1], b.shape[-1]
    if out is None:
        out = torch.empty(tuple(a.shape[:-1]) + (Ca + Cb,), dtype=torch.float32, device=a.device)
    _f32(out, *(tuple(a.shape[:-1]) + (Ca + Cb,)))
    L.check(L.load().hdrsky_concat2(_p(_f32(a)), Ca, _p(_f32(b)), Cb, a.numel() // Ca, _p(out), _stream()), "concat2")
    return out


def concat_rows4(parts, out):
    """out [M, sum w_i] = the four [M, w_i] fp32 matrices side by side (one launch; parallel.GradientExchange)."""
    M = parts[0].shape[0]
    for t in parts:
        _f32(t, M, t.shape[1])
    _f32(out, M, sum(t.shape[1] for t in parts))
    a, b, c, d = parts
    L.check(L.load().hdrsky_concat_rows4(_p(a), a.shape[1], _p(b), b.shape[1], _p(c), c.shape[1], _p(d), d.shape[1], M,
                                         _p(out), _stream()), "concat_rows4")
    return out


def vgg_pre(x):
    y = torch.empty_like(x)
    L.check(L.load().hdrsky_vgg_pre(_p(_f32(x)), x.numel(), _p(y), _stream()), "vgg_pre")
    return y


def flip_rgb(x):
    """Channel reversal of a [...,3] tensor (tf_utils.py:85-93)."""
    _f32(x)
    if x.shape[-1] != 3:
        raise ValueError("3-channel images expected")
    y = torch.empty_like(x)
    L.check(L.load().hdrsky_flip_rgb(_p(x), x.numel() // 3, _p(y), _stream()), "flip_rgb")
    return y


def axpby(a, sa, b=None, sb=0.0, out=None):
    y = out if out is not None else torch.empty_like(a)
    L.check(L.load().hdrsky_axpby(_p(_f32(a)), sa, _p(b), sb, a.numel(), _p(y), _stream()), "axpby")
    return y


def fc_wgrad(x, dy, dw, db, accumulate=False):
    M, Kd = x.shape
    N = dy.shape[1]
    _f32(x); _f32(dy, M, N); _f32(dw, Kd, N)
    for i in range(0, M, FC_MAX_ROWS):
        m = min(FC_MAX_ROWS, M - i)
        L.check(L.load().hdrsky_fc_wgrad(_p(x[i:i + m]), _p(dy[i:i + m]), m, Kd, N, int(accumulate or i > 0), _p(dw), _p(db),
                                         _stream()), "fc_wgrad")


def _rows2d(t, name):
    """[M, C] fp32 view whose rows may be strided (a column block of a wider buffer): (pointer, row stride in floats)."""
    if not (torch.is_tensor(t) and t.is_cuda) or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or (t.stride(0) & 3) or (t.data_ptr() & 15):
        raise ValueError("%s: fp32 [M, C] with unit column stride, a row stride that is a multiple of 4 and a 16-byte "
                         "aligned base expected" % name)
    return _p(t), t.stride(0)


def _xtdy_ws(M, Kd, N, device):
    n = L.load().hdrsky_fc_xtdy_ws_bytes(M, Kd, N)
    if n == 0:
        raise ValueError("fc x^T dy: bad shape M=%d K=%d N=%d" % (M, Kd, N))
    return torch.empty(n, dtype=torch.uint8, device=device)


def fc_xtdy_supported(Kd, N):
    """Shapes hdrsky_fc_wgrad_bf16 / hdrsky_rmsprop_fc_fused take (csrc/fc_update.hip shapes_ok: 32-row k tiles, 256-column
    workgroups).  im_height * im_width is only guaranteed to be a multiple of 64 (both multiples of 8): callers fall back to
    hdrsky_fc_wgrad + hdrsky_rmsprop_fc for the other sizes."""
    return Kd % 32 == 0 and N % 256 == 0 and Kd * N < (1 << 30)


def fc_wgrad_bf16(x, dy, dw, db, accumulate=False):
    """Dense weight / bias gradient on the matrix cores (bf16 operands, fp32 accumulation) - any number of rows."""
    (M, Kd), N = x.shape, dy.shape[1]
    _f32(dw, Kd, N)
    px, ldx = _rows2d(x, "fc_wgrad_bf16 x"); pd, ldy = _rows2d(dy, "fc_wgrad_bf16 dy")
    if dy.shape[0] != M:
        raise ValueError("fc_wgrad_bf16: x and dy disagree on the number of rows")
    ws = _xtdy_ws(M, Kd, N, x.device)
    L.check(L.load().hdrsky_fc_wgrad_bf16(px, ldx, pd, ldy, M, Kd, N, int(accumulate), _p(dw), _p(db), _p(ws), _stream()),
            "fc_wgrad_bf16")


def rmsprop_fc_fused(w, ms, x, dy, pf, lr, db=None, rho=0.9, eps=1e-7, gscale=1.0, bias=None, bias_ms=None):
    """RMSprop of a Dense kernel with its gradient gscale * x^T dy recomputed inside the update (never materialised);
    refreshes the bf16 images of `pf`; db (optional) receives the bias gradient; bias / bias_ms [N] (with db): the bias vector
    gets its RMSprop step inside the same call."""
    (M, Kd), N = x.shape, dy.shape[1]
    _f32(w, Kd, N); _f32(ms, Kd, N)
    if pf.pk_lo is not None or (pf.K, pf.N) != (Kd, N) or dy.shape[0] != M:
        raise ValueError("rmsprop_fc_fused: BF16 images of the same kernel and matching operand rows only")
    px, ldx = _rows2d(x, "rmsprop_fc_fused x"); pd, ldy = _rows2d(dy, "rmsprop_fc_fused dy")
    ws = _xtdy_ws(M, Kd, N, x.device)
    if bias is not None:
        _f32(bias, N); _f32(bias_ms, N); _f32(db, N)
        L.check(L.load().hdrsky_rmsprop_fc_fused_bias(_p(w), _p(ms), px, ldx, pd, ldy, M, Kd, N, lr, rho, eps, gscale, _p(pf.pk_hi),
                                                      _p(pf.nat_hi), _p(db), _p(bias), _p(bias_ms), _p(ws), _stream()),
                "rmsprop_fc_fused_bias")
        return
    L.check(L.load().hdrsky_rmsprop_fc_fused(_p(w), _p(ms), px, ldx, pd, ldy, M, Kd, N, lr, rho, eps, gscale, _p(pf.pk_hi),
                                             _p(pf.nat_hi), _p(db), _p(ws), _stream()), "rmsprop_fc_fused")


def rmsprop_fc_fused_prepare(x, dy, Kd, N, lr, ws, db, rho=0.9, eps=1e-7, gscale=1.0, bias=None, bias_ms=None):
    """First half of rmsprop_fc_fused (hdrsky_rmsprop_fc_fused_prepare): the operands' bf16 images into the caller's workspace ws
    (fc_xtdy_ws), the bias gradient db and - given bias / bias_ms - the bias vector's step.  rmsprop_fc_fused_apply(…, ws) is the update."""
    M = x.shape[0]
    if tuple(x.shape) != (M, Kd) or tuple(dy.shape) != (M, N):
        raise ValueError("rmsprop_fc_fused_prepare: operand shapes")
    px, ldx = _rows2d(x, "rmsprop_fc_fused x"); pd, ldy = _rows2d(dy, "rmsprop_fc_fused dy")
    if ws.numel() < L.load().hdrsky_fc_xtdy_ws_bytes(M, Kd, N):
        raise ValueError("rmsprop_fc_fused_prepare: workspace too small")
    _f32(db, N)
    if bias is not None:
        _f32(bias, N); _f32(bias_ms, N)
    L.check(L.load().hdrsky_rmsprop_fc_fused_prepare(px, ldx, pd, ldy, M, Kd, N, lr, rho, eps, gscale, _p(db), _p(bias), _p(bias_ms), _p(ws),
                                                     _stream()), "rmsprop_fc_fused_prepare")


def rmsprop_fc_fused_apply(w, ms, M, pf, lr, ws, rho=0.9, eps=1e-7, gscale=1.0):
    """Second half: w, ms and the bf16 images of `pf` from the operand images in ws (M = the row count _prepare was given)."""
    Kd, N = w.shape
    _f32(w, Kd, N); _f32(ms, Kd, N)
    if pf.pk_lo is not None or (pf.K, pf.N) != (Kd, N):
        raise ValueError("rmsprop_fc_fused_apply: BF16 images of the same kernel only")
    L.check(L.load().hdrsky_rmsprop_fc_fused_apply(_p(w), _p(ms), M, Kd, N, lr, rho, eps, gscale, _p(pf.pk_hi), _p(pf.nat_hi), _p(ws), _stream()),
            "rmsprop_fc_fused_apply")


def fc_xtdy_ws(M, Kd, N, device):
    """Workspace of the fused Dense update for M operand rows (persistent when the update is deferred)."""
    return _xtdy_ws(M, Kd, N, device)


def rmsprop(w, g, ms, lr, rho=0.9, eps=1e-7, gscale=1.0):
    n = w.numel()
    _f32(w); _f32(g, n); _f32(ms, n)
    L.check(L.load().hdrsky_rmsprop(_p(w), _p(g), _p(ms), n, lr, rho, eps, gscale, _stream()), "rmsprop")


def rmsprop2(w1, g1, ms1, w2, g2, ms2, lr, rho=0.9, eps=1e-7, gscale=1.0):
    """rmsprop over two flat buffers (two optimizers with the same hyper-parameters) in one launch."""
    n1, n2 = w1.numel(), w2.numel()
    _f32(w1); _f32(g1, n1); _f32(ms1, n1); _f32(w2); _f32(g2, n2); _f32(ms2, n2)
    L.check(L.load().hdrsky_rmsprop2(_p(w1), _p(g1), _p(ms1), n1, _p(w2), _p(g2), _p(ms2), n2, lr, rho, eps, gscale, _stream()), "rmsprop2")


def rmsprop_fc(w, g, ms, pf, lr, rho=0.9, eps=1e-7, gscale=1.0):
    """RMSprop step of a Dense kernel w [K,N] (views of the flat buffers) fused with the refresh of its PackedFC images."""
    Kd, N = w.shape
    _f32(w); _f32(g, Kd, N); _f32(ms, Kd, N)
    if pf.pk_lo is not None or (pf.K, pf.N) != (Kd, N):
        raise ValueError("rmsprop_fc: BF16 images of the same kernel only")
    L.check(L.load().hdrsky_rmsprop_fc(_p(w), _p(g), _p(ms), Kd, N, lr, rho, eps, gscale, _p(pf.pk_hi), _p(pf.nat_hi),
                                       _stream()), "rmsprop_fc")


# ------------------------------------------------------------------------------------------------
# sample-resident 3x3 conv + InstanceNorm (csrc/res_conv.hip): the res-block chain on bf16 activations
# ------------------------------------------------------------------------------------------------
def _bf16(t, *shape):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise ValueError("expected a contiguous CUDA bfloat16 tensor")
    if shape and tuple(t.shape) != tuple(shape):
        raise ValueError("shape %s != expected %s" % (tuple(t.shape), tuple(shape)))
    return t


def resconv_supported(H, W, Cin, Cout, KH=3, KW=3):
    return bool(L.load().hdrsky_resconv_supported(H, W, Cin, Cout, KH, KW))


def to_bf16(x, out=None):
    """bf16 copy (round to nearest even) of a contiguous fp32 tensor: entry of a bf16 activation chain."""
    _f32(x)
    y = out if out is not None else torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _bf16(y, *x.shape)
    L.check(L.load().hdrsky_to_bf16(_p(x), _p(y), x.numel(), _stream()), "to_bf16")
    return y


def resconv_fwd(x, pw: PackedConv, bias, gamma, beta, slope, residual=None, want_bf16=True, want_f32=False,
                save=False, eps=IN_EPS):
    """One half of generator.resBlock.call (generator.py:26-35) in one launch:
    y = leaky(InstanceNorm(conv3x3(x) + bias), slope) [+ residual].  x: bf16 [B,8,32,Cin] final activations.
    Returns dict(bf16=, f32=, xhat=, inv=): the output as bf16 / fp32 and - save=True - what the backward launch
    re-reads (bf16 normalised pre-activation, rstd [B,Cout])."""
    B, H, W, Cin = x.shape
    _bf16(x)
    Cout = pw.Cout
    if pw.Cin != Cin or (pw.KH, pw.KW) != (3, 3) or pw.flip or not resconv_supported(H, W, Cin, Cout):
        raise ValueError("resconv_fwd: unsupported layer")
    a = L.ResconvArgs()
    a.B, a.Cin, a.Cout, a.mode, a.slope, a.eps = B, Cin, Cout, L.RC_FWD, float(slope), float(eps)
    out = {}
    if want_bf16:
        out["bf16"] = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=x.device)
    if want_f32:
        out["f32"] = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    if save:
        out["xhat"] = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=x.device)
        out["inv"] = torch.empty((B, Cout), dtype=torch.float32, device=x.device)
    if residual is not None:
        _f32(residual, B, H, W, Cout)
    if bias is not None:
        _f32(bias, Cout)
    a.x, a.w, a.bias, a.gamma, a.beta, a.res = _p(x), _p(pw.hi), _p(bias), _p(_f32(gamma, Cout)), _p(_f32(beta, Cout)), _p(residual)
    a.y_bf16, a.y_f32, a.xhat_out, a.inv_out = _p(out.get("bf16")), _p(out.get("f32")), _p(out.get("xhat")), _p(out.get("inv"))
    L.check(L.load().hdrsky_resconv(a, _stream()), "resconv (fwd)")
    if TRACE is not None:
        _trace("resconv", "resconv_kernel<%d>" % (Cin // 32), "3x3 %d->%d @%dx%d B=%d + InstanceNorm fwd" % (Cin, Cout, H, W, B),
               2.0 * B * H * W * 9 * Cin * Cout,
               lambda a_=a, k_=(x, pw, bias, gamma, beta, residual, out): L.check(L.load().hdrsky_resconv(a_, _stream()), "resconv"),
               default_label="gen.res.* (sample-resident half block)")
    return out


def resconv_bwd(dy, pwT, skip=None, norm=None, want_f32=False, want_bf16=True, shape=None):
    """Data gradient of a 3x3 conv on the 8x32 maps fused with what follows it in the backward chain:
    g = conv3x3(dy; flipped filter pwT) [+ skip]                (fp32, returned when want_f32: the residual stream's gradient)
    norm = dict(xhat=, inv=, gamma=, beta=, slope=, dgb=): then the gradient through leaky(InstanceNorm(.)) whose forward
    launch saved xhat / inv:  dc = gamma*inv*(dz - mean(dz) - xhat*mean(dz*xhat)), dz = g*leaky'(gamma*xhat+beta), written
    as bf16 (the next data gradient's operand) and dgb[B,2,C] receives the per-sample (d gamma, d beta) terms.
    dy=None (then skip is required): no convolution - the norm backward of `skip` alone (entry of the chain)."""
    if dy is not None:
        _bf16(dy)
        B, H, W, Cin = dy.shape
        if pwT.Cin != Cin or (pwT.KH, pwT.KW) != (3, 3) or not pwT.flip:
            raise ValueError("resconv_bwd: needs the transpose_flip image of a 3x3 filter matching dy")
        Cout = pwT.Cout
    else:
        B, H, W, Cout = skip.shape
        Cin = Cout
    if not resconv_supported(H, W, Cin, Cout):
        raise ValueError("resconv_bwd: unsupported layer")
    a = L.ResconvArgs()
    a.B, a.Cin, a.Cout, a.mode, a.eps = B, Cin, Cout, L.RC_BWD, IN_EPS
    a.slope = float(norm["slope"]) if norm else 1.0
    out = {}
    if want_bf16:
        out["bf16"] = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=(dy if dy is not None else skip).device)
    if want_f32:
        out["f32"] = torch.empty((B, H, W, Cout), dtype=torch.float32, device=(dy if dy is not None else skip).device)
    if skip is not None:
        _f32(skip, B, H, W, Cout)
    a.x, a.w, a.res = _p(dy), _p(pwT.hi) if dy is not None else None, _p(skip)
    if norm:
        _bf16(norm["xhat"], B, H, W, Cout); _f32(norm["inv"], B, Cout)
        a.xhat_in, a.inv_in = _p(norm["xhat"]), _p(norm["inv"])
        a.gamma, a.beta = _p(_f32(norm["gamma"], Cout)), _p(_f32(norm["beta"], Cout))
        if norm.get("dgb") is not None:
            a.dgb = _p(_f32(norm["dgb"], B, 2, Cout))
    a.y_bf16, a.y_f32 = _p(out.get("bf16")), _p(out.get("f32"))
    L.check(L.load().hdrsky_resconv(a, _stream()), "resconv (bwd)")
    if TRACE is not None and dy is not None:
        _trace("resconv", "resconv_kernel<%d>" % (Cin // 32), "3x3 %d->%d @%dx%d B=%d data gradient + InstanceNorm bwd" % (Cin, Cout, H, W, B),
               2.0 * B * H * W * 9 * Cin * Cout,
               lambda a_=a, k_=(dy, pwT, skip, norm, out): L.check(L.load().hdrsky_resconv(a_, _stream()), "resconv"),
               default_label="gen.res.* (data gradient + InstanceNorm backward)")
    return out


class DgbReducer:
    """One-launch, fixed-order reduction over the batch of several per-sample tables (hdrsky_dgb_reduce):
    entries = [(part [B,2,C], dst0 [C] or None, dst1 [C] or None), ...]; dst_k += sum_b part[b, k].  resconv's `dgb` holds
    (d gamma, d beta), norm_act_bwd's `sums` (d beta, d gamma).  The pointer table is uploaded in the constructor (not
    capturable); run() is one launch."""

    def __init__(self, entries):
        self.B = entries[0][0].shape[0]
        rows = []
        for part, d0, d1 in entries:
            C = part.shape[2]
            _f32(part, self.B, 2, C)
            for d in (d0, d1):
                if d is not None:
                    _f32(d, C)
            rows.append([part.data_ptr(), d0.data_ptr() if d0 is not None else 0, d1.data_ptr() if d1 is not None else 0, C])
        self.n = len(rows)
        self.table = torch.tensor(rows, dtype=torch.int64, device=entries[0][0].device)
        self._keep = entries

    def run(self):
        L.check(L.load().hdrsky_dgb_reduce(_p(self.table), self.n, self.B, _stream()), "dgb_reduce")


# ------------------------------------------------------------------------------------------------
# distortion-aware convolution (csrc/da_conv.hip)
# ------------------------------------------------------------------------------------------------
def adam(w, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-7, gscale=1.0):
    """One tf.keras Adam step (OptimizerV2 form) on flat buffers; `step` is the 1-based update count."""
    n = w.numel()
    _f32(w); _f32(g); _f32(m); _f32(v)
    import numpy as np                      # lr_t in float32, as OptimizerV2 forms it
    f = np.float32
    lr_t = float(f(lr) * np.sqrt(f(1) - np.power(f(beta2), f(step))) / (f(1) - np.power(f(beta1), f(step))))
    L.check(L.load().hdrsky_adam(_p(w), _p(g), _p(m), _p(v), n, lr_t, beta1, beta2, eps, gscale, _stream()), "adam")


def da_offsets(h, w, ksize=3, dilation_rate=1, skydome=True):
    """Host float32 offset table [h, k*k, 2] (distortion_aware_ops.py:198-270)."""
    import ctypes
    import numpy as np
    out = np.zeros((h, ksize * ksize, 2), np.float32)
    L.check(L.load().hdrsky_da_offsets(h, w, ksize, dilation_rate, int(skydome), out.ctypes.data_as(ctypes.c_void_p)),
            "da_offsets")
    return out


def _da_host_table(h, w, ksize, dilation_rate, skydome):
    """hdrsky_da_sample_table of the geometry: (idx, wt) [h*w, k*k, 4] numpy arrays (the forward's corners / weights)."""
    import ctypes
    import numpy as np
    k2 = ksize * ksize
    offs = da_offsets(h, w, ksize, dilation_rate, skydome)
    idx = np.zeros((h * w, k2, 4), np.int32)
    wt = np.zeros((h * w, k2, 4), np.float32)
    L.check(L.load().hdrsky_da_sample_table(offs.ctypes.data_as(ctypes.c_void_p), h, w, ksize,
                                            idx.ctypes.data_as(ctypes.c_void_p), wt.ctypes.data_as(ctypes.c_void_p)),
            "da_sample_table")
    return offs, idx, wt


DA_GROUPS = (1, 2, 4, 8, 16)   # tiles per workgroup the region kernels may choose from


def da_row_lo(idx, w):
    """Source rows of groups of 1, 2, 4, 8, 16 consecutive 64-pixel tiles (row-major) of a sample table idx [h*w, k*k, KM]
    (pixel indices, < 0 = none): (row_lo int32 [5, tiles], spans int32 [5]) - group g of level l reads rows
    row_lo[l, g] .. row_lo[l, g] + spans[l] - 1 at most.  What hdrsky_da_conv2d_fwd / _dgrad need to stage a group's source
    rows in LDS."""
    import numpy as np
    hw = idx.shape[0]
    nt = (hw + 63) // 64
    rows = np.where(idx >= 0, idx // w, -1).reshape(hw, -1)
    big = np.iinfo(np.int32).max
    lo_all = np.zeros((len(DA_GROUPS), nt), np.int32)
    spans = np.zeros(len(DA_GROUPS), np.int32)
    for l, G in enumerate(DA_GROUPS):
        ng = (nt + G - 1) // G
        pad = ng * G * 64 - hw
        r = np.concatenate([rows, np.full((pad, rows.shape[1]), -1, rows.dtype)], 0) if pad else rows
        r = r.reshape(ng, -1)
        hi = r.max(1)
        lo = np.where(hi >= 0, np.where(r >= 0, r, big).min(1), 0)
        lo_all[l, :ng] = lo
        spans[l] = int(np.where(hi >= 0, hi - lo + 1, 1).max())
    return lo_all, spans


_DA_OFFS = {}


def da_offsets_device(h, w, ksize=3, dilation_rate=1, skydome=True, device="cuda"):
    """Device copy of da_offsets(...) for da_conv2d, carrying the source-row table of its 64-pixel tiles (attribute
    `da_rows` = (row_lo device int32 [5, tiles], spans host int32 [5]: kernels.da_row_lo) so that the forward can stage the
    source rows of a group of tiles in LDS."""
    key = (h, w, ksize, dilation_rate, bool(skydome), str(device))
    if key not in _DA_OFFS:
        offs, idx, _ = _da_host_table(h, w, ksize, dilation_rate, skydome)
        lo, spans = da_row_lo(idx, w)
        t = torch.from_numpy(offs).to(device)
        t.da_rows = (torch.from_numpy(lo).to(device), spans)
        _DA_OFFS[key] = t
    return _DA_OFFS[key]


def _da_rows(t):
    """(device pointer of row_lo, host pointer of spans) of a tensor made by da_offsets_device / da_transpose_table."""
    r = getattr(t, "da_rows", None)
    return (r[0].data_ptr(), r[1].ctypes.data) if r is not None else (None, None)


def da_materialised(compute):
    """Single-product mode: a distortion-aware layer runs as the reference writes it (distortion_aware_ops.py:107-121) - the
    gathered operand G [B,H,W,k*k*C] is written once as bf16 (hdrsky_da_gather_bf16) and the generic 1x1 conv / weight-gradient
    kernels run on it - instead of the kernels that gather while staging (csrc/da_conv.hip), which stay for the split-product
    mode.  The fused kernels re-gather per output-channel block and sit at 1-2 % of the matrix-core peak on the 128x512 maps
    (profiles/r04_da_mat_ab.txt).  HDRSKY_DA_MAT=0: the fused kernels (switch)."""
    return compute == BF16 and HOOKS.H.da_mat


DA_MAT_MIN_PIXELS = {"fwd": 1024, "dgrad": 1024, "wgrad": 1024}


def da_mat_ok(compute, ksize, C, pixels=None, what="wgrad"):
    """... for one launch of a layer whose matmul runs over k*k*C channels on maps of `pixels` = H*W.  Channels: where the generic
    conv could take them in a few groups (power-of-two divisors of the channel count: 9 x 32, 9 x 64, 9 x 128 - not the 49 x 32 of
    a 7 x 7 layer; hdrsky_gemm1x1_bf16 takes every multiple of 64).  Size: from 1024 pixels per sample, for all three launches
    (profiles/r04_da_mat_ab.txt, microbench_da_mat.py; us alone, written incl. its gather / fused): 128->128 on 32x128 maps at batch 8:
    forward 44 / 63, data gradient 56 / 79, kernel gradient 89 / 212; 64->32 on 128x512: 300 / 499, 402 / 432, 230 / 1024;
    64->64 on 16x64 at batch 32: 27 / 32, 31 / 40, 37 / 58.  On the 8x32 maps of the 32x128 network the fused kernels - a workgroup
    stages a sample's few source rows once - are level or ahead (25 / 22, 30 / 20; the kernel gradients of its twelve res-block layers
    share ONE fused launch), so they stay there."""
    kc = ksize * ksize * C
    g = kc & -kc
    return da_materialised(compute) and g >= 32 and kc // g <= 16 and (pixels is None or pixels >= DA_MAT_MIN_PIXELS[what])


def da_gather_bf16(x, offs=None, table=None, ksize=3):
    """G (bfloat16) [B,H,W,k*k*C]: the gathered operand from the forward's corners (offs) or from a sample table (gidx, gw)
    [H*W, k*k, km] - da_transpose_table's makes G the operand of the data gradient (x = dY).  x: fp32 or bfloat16."""
    B, H, W, C = x.shape
    x16 = _raw(x)
    k2 = ksize * ksize
    G = torch.empty((B, H, W, k2 * C), dtype=torch.bfloat16, device=x.device)
    if offs is not None:
        _f32(offs, H, k2, 2)
        gidx = gw = None; km = 0
    else:
        gidx, gw = table
        km = gidx.shape[-1]
        if tuple(gidx.shape) != (H * W, k2, km) or gidx.dtype != torch.int32 or tuple(gw.shape) != tuple(gidx.shape):
            raise ValueError("da_gather_bf16: table does not match the map")
        _f32(gw)
    L.check(L.load().hdrsky_da_gather_bf16(_p(x), x16, _p(offs), _p(gidx), _p(gw), km, B, H, W, C, ksize, _p(G), _stream()),
            "da_gather_bf16")
    return G


def gemm1x1(G, pw1: PackedConv, bias=None, want_stats=False, out_bf16=False):
    """y [B,H,W,N] = G [B,H,W,K] (bfloat16, final) x the 1x1 view of a packed filter (PackedConv.as_1x1()) + bias, single-product
    mode: hdrsky_gemm1x1_bf16 where it takes the shape (whole 128-pixel tiles per sample, K % 64 == 0, N % 32 == 0), the generic
    conv as a 1x1 layer otherwise.  Returns (y, Stats | None)."""
    _bf16(G)
    B, H, W, Kc = G.shape
    N = pw1.Cout
    lib = L.load()
    if (pw1.KH, pw1.KW, pw1.Cin) != (1, 1, Kc):
        raise ValueError("gemm1x1: filter view does not match the operand")
    if not lib.hdrsky_gemm1x1_supported(H * W, Kc, N):
        return conv2d(G, pw1, bias, compute=BF16, want_stats=want_stats, out_bf16=out_bf16)
    if bias is not None:
        _f32(bias, N)
    y = torch.empty((B, H, W, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=G.device)
    st = None
    if want_stats:
        nparts = lib.hdrsky_gemm1x1_stats_nparts(H * W)
        st = Stats(torch.empty((B, nparts, 2, N), dtype=torch.float32, device=G.device), nparts, H * W)
    L.check(lib.hdrsky_gemm1x1_bf16(_p(G), _p(pw1.hi), _p(bias), B, H * W, Kc, N, _p(y), int(out_bf16), _p(st.part) if st else None,
                                    _stream()), "gemm1x1_bf16")
    if TRACE is not None:
        _trace("conv", "gemm1x1_kernel", "1x1 %d->%d @%dx%d B=%d (written gathered operand)" % (Kc, N, H, W, B), 2.0 * B * H * W * Kc * N,
               lambda: lib.hdrsky_gemm1x1_bf16(_p(G), _p(pw1.hi), _p(bias), B, H * W, Kc, N, _p(y), int(out_bf16), _p(st.part) if st else None, _stream()))
    return y, st


def da_conv2d(x, pw: PackedConv, bias, offs, compute=BF16, want_stats=False, train=False, reuse_operand=False,
              operand: Optional["Operand"] = None):
    """distortion_aware_ops.conv2d.call: offs = device tensor [H, k*k, 2] from da_offsets(H, W, k, ...).
    want_stats: also return the InstanceNorm partials of y (Stats, as conv2d does) -> (y, Stats).
    train: the layer's kernel gradient will follow (kept for callers - the written operand is the faster forward from 1024
    pixels per sample with or without it: da_mat_ok).  operand (an Operand handle, training): where the layer runs on its
    written gathered operand, the handle receives it and da_wgrad_job(operand=) reads it again; without a handle nothing is
    kept.  reuse_operand: a second layer on the SAME, unchanged input (the decoders' first deconvolutions) takes the operand
    the handle already holds for this (offsets, k, input) - the caller's promise that x was not rewritten in between."""
    B, H, W, C = x.shape
    if C != pw.Cin or pw.KH != pw.KW:
        raise ValueError("filter / input mismatch")
    if da_mat_ok(compute, pw.KH, C, H * W, "fwd"):
        key = _da_key(x, offs, pw.KH)
        if reuse_operand and operand is not None and operand.tensor is not None and operand.key == key:
            G = operand.tensor
        else:
            G = da_gather_bf16(x, offs, ksize=pw.KH)
        if operand is not None:
            operand.tensor, operand.key = G, key      # the weight gradient of the layer reads it again (da_wgrad_job(operand=))
        y, st = gemm1x1(G, pw.as_1x1(), bias, want_stats=want_stats)
        return (y, st) if want_stats else y
    _f32(x)
    _f32(offs, H, pw.KH * pw.KW, 2)
    if bias is not None:
        _f32(bias, pw.Cout)
    if compute == BF16X3 and pw.lo is None:
        raise ValueError("BF16X3 needs the lo weight plane")
    y = torch.empty((B, H, W, pw.Cout), dtype=torch.float32, device=x.device)
    lib = L.load()
    st = None
    if want_stats:
        nparts = lib.hdrsky_da_conv_stats_nparts(H, W)
        st = Stats(torch.empty((B, nparts, 2, pw.Cout), dtype=torch.float32, device=x.device), nparts, H * W)
    row_lo, spans = _da_rows(offs)
    L.check(lib.hdrsky_da_conv2d_fwd(_p(x), _p(pw.hi), _p(pw.lo), _p(bias), _p(offs), row_lo, spans, B, H, W, C, pw.Cout, pw.KH,
                                     compute, _p(y), _p(st.part) if st else None, _stream()), "da_conv2d_fwd")
    return (y, st) if want_stats else y


def da_gather(x, offs, ksize):
    """G [B,H,W,k*k*C]: the bilinear-gathered operand of the distortion-aware conv's matmul."""
    B, H, W, C = x.shape
    _f32(x); _f32(offs, H, ksize * ksize, 2)
    G = torch.empty((B, H, W, ksize * ksize * C), dtype=torch.float32, device=x.device)
    L.check(L.load().hdrsky_da_gather(_p(x), _p(offs), B, H, W, C, ksize, _p(G), _stream()), "da_gather")
    return G


DA_KMAX = 8   # source pixels per (target pixel, tap) the data-gradient kernel gathers
_DA_TT = {}


def da_transpose_table(h, w, ksize=3, dilation_rate=1, skydome=True, device="cuda"):
    """Transposed sample table of the distortion-aware conv on an h x w map, for da_conv2d_dgrad: (gidx, gw) device tensors
    [h*w, k*k, DA_KMAX] - for target pixel q and tap slot s (the tap order of the transpose_flip filter image: slot s =
    forward tap k*k-1-s) the forward samples (p, tap) that read q and their weights.  Built once per geometry on the host
    from hdrsky_da_sample_table (the forward's own float32 arithmetic).  Returns None when some (q, tap) has more than
    DA_KMAX readers (not the case for the 3x3 layers of the model)."""
    import ctypes
    import numpy as np
    key = (h, w, ksize, dilation_rate, bool(skydome), str(device))
    if key not in _DA_TT:
        k2 = ksize * ksize
        _, idx, wt = _da_host_table(h, w, ksize, dilation_rate, skydome)
        p, t, c = np.nonzero((idx >= 0) & (wt != 0.0))
        q = idx[p, t, c]
        slot = k2 - 1 - t
        order = np.lexsort((c, p, slot, q))                 # fixed order inside each (q, slot) list: deterministic sums
        q, slot, p, wv = q[order], slot[order], p[order], wt[p, t, c][order]
        grp = q.astype(np.int64) * k2 + slot
        start = np.r_[0, np.flatnonzero(np.diff(grp)) + 1]
        pos = np.arange(grp.size) - np.repeat(start, np.diff(np.r_[start, grp.size]))
        if pos.size and int(pos.max()) >= DA_KMAX:
            _DA_TT[key] = None
        else:
            gidx = np.full((h * w, k2, DA_KMAX), -1, np.int32)
            gw = np.zeros((h * w, k2, DA_KMAX), np.float32)
            gidx[q, slot, pos] = p
            gw[q, slot, pos] = wv
            lo, spans = da_row_lo(gidx, w)
            tg = torch.from_numpy(gidx).to(device)
            tg.da_rows = (torch.from_numpy(lo).to(device), spans)    # source rows (of dy) of every 64-pixel tile
            _DA_TT[key] = (tg, torch.from_numpy(gw).to(device))
    return _DA_TT[key]


def da_conv2d_dgrad(dy, pwT: PackedConv, table, ksize, compute=BF16):
    """dx of y = da_conv2d(x; kernel): deterministic, no k*k-fold tensor.  pwT = PackedConv(kernel.view(k,k,C,F),
    transpose_flip=True); table = da_transpose_table(H, W, k, ...)."""
    B, H, W, F = dy.shape
    gidx, gw = table
    if pwT.Cin != F or not pwT.flip or (pwT.KH, pwT.KW) != (ksize, ksize):
        raise ValueError("da_conv2d_dgrad: needs the transpose_flip image of the k x k filter")
    if tuple(gidx.shape) != (H * W, ksize * ksize, DA_KMAX) or gidx.dtype != torch.int32:
        raise ValueError("da_conv2d_dgrad: table does not match the map")
    if da_mat_ok(compute, ksize, F, H * W, "dgrad"):      # the same sums as a gather of dY on the transposed table + the 1x1 conv on it
        return gemm1x1(da_gather_bf16(dy, table=table, ksize=ksize), pwT.as_1x1())[0]
    _f32(dy)
    if compute == BF16X3 and pwT.lo is None:
        raise ValueError("BF16X3 needs the lo weight plane")
    dx = torch.empty((B, H, W, pwT.Cout), dtype=torch.float32, device=dy.device)
    row_lo, spans = _da_rows(gidx)
    L.check(L.load().hdrsky_da_conv2d_dgrad(_p(dy), _p(pwT.hi), _p(pwT.lo), _p(gidx), _p(_f32(gw)), row_lo, spans, B, H, W, F, pwT.Cout, ksize,
                                            compute, _p(dx), _stream()), "da_conv2d_dgrad")
    return dx


def da_conv2d_bwd(x, dy, kernel, offs, ksize, compute=BF16, want_dx=True, pwT=None, dw=None, db=None, table=None, pwT3=None):
    """Gradients of y = da_conv2d(x; kernel [k*k*C, F], bias), C % 32 == 0: returns (dx or None, dkernel [k*k*C, F], dbias [F]).
    table (da_transpose_table) [+ pwT3 = PackedConv(kernel.view(k,k,C,F), transpose_flip=True)]: dx by the deterministic
    gather-form data gradient (hdrsky_da_conv2d_dgrad).  Without a table: dG = dY W^T as a 1x1 conv (pwT =
    PackedConv(kernel.view(1,1,k*k*C,F), transpose_flip=True)) + bilinear scatter with fp32 atomics.
    dw [k*k*C, F] / db [F]: gradients are ADDED to these instead of freshly allocated ones."""
    B, H, W, C = x.shape
    F = dy.shape[-1]
    k2 = ksize * ksize
    _f32(dy, B, H, W, F); _f32(kernel, k2 * C, F)
    # dW = G^T dY with the gather recomputed inside the weight-gradient launch (hdrsky_wgrad_job.da_*): G never exists
    if dw is None:
        dw = zero_(torch.empty((k2 * C, F), dtype=torch.float32, device=x.device))
    if db is None:
        db = zero_(torch.empty((F,), dtype=torch.float32, device=x.device))
    conv2d_wgrad_multi([da_wgrad_job(x, dy, ksize, offs, dw, db, compute)])
    dw4 = dw
    dx = None
    if want_dx and table is not None:
        # the transpose of the gather as a gather (da_conv2d_dgrad): deterministic, nothing k*k-fold in memory
        if pwT3 is None:
            pwT3 = PackedConv(kernel.view(ksize, ksize, C, F), precise=(compute == BF16X3), transpose_flip=True)
        dx = da_conv2d_dgrad(dy, pwT3, table, ksize, compute)
    elif want_dx:
        if pwT is None:   # dG = dY W^T as a 1x1 conv: the transpose_flip image of the kernel viewed as a 1x1 filter
            pwT = PackedConv(kernel.view(1, 1, k2 * C, F), precise=(compute == BF16X3), transpose_flip=True)
        dG, _ = conv2d(dy, pwT, None, compute=compute)
        dx = zero_(torch.empty_like(x))
        L.check(L.load().hdrsky_da_scatter(_p(dG), _p(offs), B, H, W, C, ksize, _p(dx), _stream()), "da_scatter")
    return dx, dw4.view(k2 * C, F), db


# ------------------------------------------------------------------------------------------------
# device-side input synthesis (train.py:42-94)
# ------------------------------------------------------------------------------------------------
def ldr_synth(hdr, t, sigma_s, sigma_c, noise_s, noise_c, crf):
    """train.py:54-94 without the JPEG round trip -> (hdr_t, jpeg_img_float stand-in), both [B,H,W,3]."""
    B, H, W, C = hdr.shape
    if C != 3:
        raise ValueError("3-channel images expected")
    _f32(hdr); _f32(t, B); _f32(sigma_s, B, 3); _f32(sigma_c, B, 3); _f32(noise_s, B, H, W, 3); _f32(noise_c, B, H, W, 3)
    _f32(crf); 
    if crf.shape[0] != B:
        raise ValueError("one response curve per sample")
    hdr_t, ldr = torch.empty_like(hdr), torch.empty_like(hdr)
    L.check(L.load().hdrsky_ldr_synth(_p(hdr), _p(t), _p(sigma_s), _p(sigma_c), _p(noise_s), _p(noise_c), _p(crf),
                                      crf.shape[1], B, H, W, _p(hdr_t), _p(ldr), _stream()), "ldr_synth")
    return hdr_t, ldr


def batch_jpeg_qualities(B):
    """train.py:89: sample i of a batch of B is re-compressed at quality int(round(i / (B - 1) * 10 + 90))."""
    return [int(round(float(i) / float(B - 1) * 10.0 + 90.0)) if B > 1 else 90 for i in range(B)]


_JPEG_Q = {}


def jpeg_roundtrip(ldr, quality=None, order="bgr", out=None):
    """train.py:86-92: tf.image.adjust_jpeg_quality on the 8-bit image of every sample (libjpeg encode + decode, bit
    exact, entropy coding skipped).  ldr [B,H,W,3] float holding k/255; quality: per-sample list / int tensor (default:
    the reference's 90..100 ramp over the batch); order: which channel is red ("rgb" or "bgr")."""
    B, H, W, C = ldr.shape
    _f32(ldr)
    if C != 3 or order not in ("rgb", "bgr"):
        raise ValueError("jpeg_roundtrip: [B,H,W,3] images, order 'rgb' or 'bgr'")
    if quality is None:
        key = (B, ldr.device)
        if key not in _JPEG_Q:
            _JPEG_Q[key] = torch.tensor(batch_jpeg_qualities(B), dtype=torch.int32, device=ldr.device)
        quality = _JPEG_Q[key]
    elif not torch.is_tensor(quality):
        quality = torch.tensor([int(q) for q in quality], dtype=torch.int32, device=ldr.device)
    if quality.dtype != torch.int32 or quality.numel() != B or not quality.is_cuda:
        raise ValueError("jpeg_roundtrip: one int32 quality per sample on the device")
    lib = L.load()
    ws = torch.empty(int(lib.hdrsky_jpeg_roundtrip_ws_bytes(B, H, W)), dtype=torch.uint8, device=ldr.device)
    out = torch.empty_like(ldr) if out is None else out
    L.check(lib.hdrsky_jpeg_roundtrip(_p(ldr), _p(quality), B, H, W, 1 if order == "bgr" else 0, _p(ws), _p(out), _stream()),
            "jpeg_roundtrip")
    return out


def vmf_target(elevation, azimuth, H, W, kappa=80.0):
    """train.py:42-52: von-Mises-Fisher pmf over the H*W sky bins for each sample's sun elevation (row units)."""
    B = elevation.numel()
    _f32(elevation, B)
    out = torch.empty((B, H * W), dtype=torch.float32, device=elevation.device)
    L.check(L.load().hdrsky_vmf_target(_p(elevation), float(azimuth), B, H, W, float(kappa), _p(out), _stream()), "vmf_target")
    return out
