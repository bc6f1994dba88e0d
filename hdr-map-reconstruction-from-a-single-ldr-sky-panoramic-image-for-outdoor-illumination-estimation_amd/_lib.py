"""ctypes binding of libhdrsky.so (the C ABI declared in include/hdrsky.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing a kernel raises immediately (the GPU tests must never pass on a silent
eager/PyTorch path).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhdrsky.so")

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

HDRSKY_BF16, HDRSKY_BF16X3 = 0, 1
HDRSKY_EUNSUPPORTED = -2
IN_NONE, IN_AFFINE, IN_PARTIALS = 0, 1, 2


class ConvDesc(ctypes.Structure):
    """Mirror of ``hdrsky_conv_desc`` (include/hdrsky.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("B", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad_t", "pad_l",
                 "upsample", "dilate", "Hc", "Wc", "compute", "in_mode", "ss_bstride", "in_nparts")] + \
               [("in_eps", c_float), ("in_slope", c_float), ("out_slope", c_float),
                ("final_relu", ctypes.c_int32), ("want_stats", ctypes.c_int32), ("x_bf16", ctypes.c_int32),
                ("y_bf16", ctypes.c_int32), ("res_mode", ctypes.c_int32), ("mask_slope", c_float)]


P = c_void_p


class WgradJob(ctypes.Structure):
    """Mirror of ``hdrsky_wgrad_job`` (include/hdrsky.h)."""
    _fields_ = [("desc", ConvDesc)] + [(n, c_void_p) for n in
                                       ("x", "dy", "in_scale", "in_shift", "in_part", "in_gamma", "in_beta", "dw", "db")] + \
               [("x_bf16", ctypes.c_int32), ("dy_bf16", ctypes.c_int32), ("da_offs", c_void_p), ("da_ksize", ctypes.c_int32),
                ("da_C", ctypes.c_int32)]


class ResconvArgs(ctypes.Structure):
    """Mirror of ``hdrsky_resconv_args`` (include/hdrsky.h)."""
    _fields_ = [("B", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("slope", c_float), ("eps", c_float)] + \
               [(n, c_void_p) for n in ("x", "w", "bias", "gamma", "beta", "res", "xhat_in", "inv_in", "y_bf16", "y_f32",
                                        "xhat_out", "inv_out", "dgb")]


RC_FWD, RC_BWD = 0, 1

ABI_VERSION = 4      # HDRSKY_ABI_VERSION of the include/hdrsky.h these mirrors were written against
STRUCTS = {"hdrsky_conv_desc": ConvDesc, "hdrsky_wgrad_job": WgradJob, "hdrsky_resconv_args": ResconvArgs}

# name -> (restype, argtypes); every symbol include/hdrsky.h declares
SIGNATURES = {
    "hdrsky_version": (ctypes.c_char_p, []),
    "hdrsky_abi_version": (c_int, []),
    "hdrsky_sizeof": (c_size_t, [ctypes.c_char_p]),
    "hdrsky_hooks_reload": (c_int, []),
    "hdrsky_experiments_enabled": (c_int, []),
    "hdrsky_conv_desc_init": (c_int, [ctypes.POINTER(ConvDesc)] + [c_int] * 10),
    "hdrsky_conv_desc_init_dgrad": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc)]),
    "hdrsky_conv_packed_elems": (c_size_t, [c_int] * 4),
    "hdrsky_conv_pack_weights": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "hdrsky_conv_pack_weights_multi": (c_int, [P, c_int, c_int, P]),
    "hdrsky_conv_stats_nparts": (c_int, [ctypes.POINTER(ConvDesc)]),
    "hdrsky_conv_kernel_name": (c_int, [ctypes.POINTER(ConvDesc), ctypes.c_char_p, c_int]),
    "hdrsky_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc)] + [P] * 13),
    "hdrsky_conv2d_emit_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "hdrsky_conv2d_fwd_emit": (c_int, [ctypes.POINTER(ConvDesc)] + [P] * 14),
    "hdrsky_conv2d_fwd_pair": (c_int, [ctypes.POINTER(ConvDesc), P, c_int] + [P] * 18),
    "hdrsky_in_affine_pair": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, c_float, P, P, P]),
    "hdrsky_norm_act_bwd_pair": (c_int, [P, P, c_int, P, P, P, P, c_float, c_float, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "hdrsky_up2x_xf_bf16_pair": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, P, c_float, c_float, P, P]),
    "hdrsky_conv2d_wgrad": (c_int, [ctypes.POINTER(ConvDesc)] + [P] * 10),
    "hdrsky_conv2d_wgrad_multi": (c_int, [ctypes.POINTER(WgradJob), c_int, P]),
    "hdrsky_conv2d_wgrad_ws_bytes": (c_size_t, [ctypes.POINTER(WgradJob), c_int]),
    "hdrsky_conv2d_wgrad_multi_det": (c_int, [ctypes.POINTER(WgradJob), c_int, P, c_size_t, P]),
    "hdrsky_conv2d_wgrad_kernel_names": (c_int, [ctypes.POINTER(WgradJob), c_int, ctypes.c_char_p, c_int]),
    "hdrsky_wgrad2_eligible": (c_int, [ctypes.POINTER(WgradJob), c_int]),
    "hdrsky_norm_apply": (c_int, [P, c_int, P, c_int, P, P, c_float, c_float, P, P, P, c_int, c_int, c_int, c_int, P]),
    "hdrsky_in_affine": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_float, P, P, P]),
    "hdrsky_in_finalize": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_float, P, P, P, P, P]),
    "hdrsky_bn_eval_affine": (c_int, [P, P, P, P, c_float, c_int, P, P, P]),
    "hdrsky_norm_act_bwd": (c_int, [P, P, c_int, P, P, c_float, c_float, P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "hdrsky_norm_act_bwd_nslices": (c_int, [c_int] * 5),
    "hdrsky_norm_act_bwd_one_launch": (c_int, [c_int] * 4),
    "hdrsky_fc_pack_weights": (c_int, [P, c_int, c_int, P, P, P, P, P]),
    "hdrsky_fc_nsplit": (c_int, [c_int]),
    "hdrsky_fc_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_fc_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_fc_fwd_fin": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P, P, P, P]),
    "hdrsky_fc_dgrad_fin": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P, P, P, P]),
    "hdrsky_fc_finalize": (c_int, [P, c_int, c_int, c_int, P, c_int, P, P, P, P]),
    "hdrsky_global_max": (c_int, [P, c_size_t, P, P]),
    "hdrsky_softmax_head": (c_int, [P, c_int, c_int, c_int, P, P, P, P, P]),
    "hdrsky_softmax_head_pick": (c_int, [P, c_int, c_int, c_int, P, P, P, P, P, P, P, P]),
    "hdrsky_softmax_pick_bwd": (c_int, [P, P, P, c_int, c_int, P, P, P]),
    "hdrsky_spatial_sum": (c_int, [P, c_int, c_int, c_int, c_float, P, P]),
    "hdrsky_grad_cam": (c_int, [P, P, c_int, c_float, c_int, c_int, c_int, P, P]),
    "hdrsky_grad_cam3": (c_int, [P, P, P, P, P, P, P, c_int, P]),
    "hdrsky_plz_build": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P]),
    "hdrsky_dense_heads": (c_int, [P, P, P, c_float, c_int, c_int, c_int, P, P, c_int, P, P]),
    "hdrsky_sun_rad": (c_int, [P, P, P, c_int, P, P, c_int, c_int, P, P, P, P, P]),
    "hdrsky_blend": (c_int, [P, P, c_int, c_float, P, P, P, P, P, P]),
    "hdrsky_tonemap": (c_int, [P, P, c_size_t, c_int, P]),
    "hdrsky_leaky_relu": (c_int, [P, P, c_size_t, c_float, P]),
    "hdrsky_ldr_synth": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "hdrsky_vmf_target": (c_int, [P, c_float, c_int, c_int, c_int, c_float, P, P]),
    "hdrsky_jpeg_roundtrip_ws_bytes": (c_size_t, [c_int] * 3),
    "hdrsky_crc32c": (ctypes.c_uint32, [P, c_size_t, ctypes.c_uint32]),
    "hdrsky_jpeg_roundtrip": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "hdrsky_da_offsets": (c_int, [c_int, c_int, c_int, c_int, c_int, P]),
    "hdrsky_da_conv2d_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "hdrsky_da_conv_stats_nparts": (c_int, [c_int, c_int]),
    "hdrsky_da_gather": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_da_scatter": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_da_gather_bf16": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_gemm1x1_supported": (c_int, [c_int, c_int, c_int]),
    "hdrsky_gemm1x1_stats_nparts": (c_int, [c_int]),
    "hdrsky_gemm1x1_bf16": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, P]),
    "hdrsky_da_sample_table": (c_int, [P, c_int, c_int, c_int, P, P]),
    "hdrsky_da_conv2d_wgrad_ws_bytes": (c_size_t, [P, P, c_int, c_int, c_int, c_int, c_int, c_int]),
    "hdrsky_da_conv2d_wgrad": (c_int, [P, P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "hdrsky_da_conv2d_dgrad": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_bn_train_finalize": (c_int, [P, c_int, c_int, c_int, P, P, c_float, c_float, P, P, P, P, P, P, c_int, P]),
    "hdrsky_zero": (c_int, [P, c_size_t, P]),
    "hdrsky_bn_bwd_nblocks": (c_int, []),
    "hdrsky_bn_act_bwd": (c_int, [P, P, P, P, P, P, c_float, c_int, c_int, P, P, P, P, c_int, P]),
    "hdrsky_bn_act_bwd_reduce": (c_int, [P, P, P, P, P, P, c_float, c_int, c_int, P, P]),
    "hdrsky_bn_act_bwd_apply": (c_int, [P, P, P, P, P, P, c_float, c_int, c_int, P, c_int, ctypes.c_double, P, c_int, P, P, P, P,
                                        c_int, P]),
    "hdrsky_sun_rad_bwd_slices": (c_int, [c_int]),
    "hdrsky_sun_rad_bwd_reduce": (c_int, [P, P, P, P, P, c_int, c_int, P, P, P]),
    "hdrsky_sun_rad_bwd_apply": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P]),
    "hdrsky_affine_act_bwd": (c_int, [P, P, P, P, c_float, c_size_t, c_int, P, c_int, P]),
    "hdrsky_maxpool_fwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_maxpool_relu_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_up2x_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_up2x_bwd": (c_int, [P, c_int, c_int, c_int, c_int, c_float, c_int, P, P]),
    "hdrsky_blur3": (c_int, [P, c_int, c_int, c_int, c_int, c_float, c_int, P, P]),
    "hdrsky_dog_mid": (c_int, [P, c_int, c_int, c_int, c_int, c_float, P, P, P]),
    "hdrsky_dog_mid_bwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "hdrsky_dog_loss": (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, P, P, P]),
    "hdrsky_l1": (c_int, [P, P, c_size_t, c_float, c_float, P, P, c_int, P]),
    "hdrsky_mse": (c_int, [P, c_float, c_size_t, c_float, c_float, P, P, P]),
    "hdrsky_kl": (c_int, [P, P, c_int, c_int, P, P, P]),
    "hdrsky_softmax_bwd": (c_int, [P, P, P, c_int, c_int, P, P]),
    "hdrsky_blend_bwd": (c_int, [P, P, P, P, c_size_t, P, P, P]),
    "hdrsky_head_bwd": (c_int, [P] * 9 + [c_size_t, P, P, P, P]),
    "hdrsky_decoder_tail_bwd": (c_int, [P, P, P, c_size_t, P, P, P]),
    "hdrsky_sun_rad_bwd": (c_int, [P, P, P, P, P, c_int, c_int, P, P, P, P]),
    "hdrsky_dense_heads_bwd": (c_int, [P, P, P, c_float, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P]),
    "hdrsky_slice_channels": (c_int, [P, c_size_t, c_int, c_int, c_int, c_float, c_int, P, P]),
    "hdrsky_pad_channels": (c_int, [P, c_size_t, c_int, c_int, P, P]),
    "hdrsky_maxpool_fwd_bf16": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P]),
    "hdrsky_maxpool_relu_bwd_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P]),
    "hdrsky_maxpool_relu_l1_bwd_bf16": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P, P, c_int, P]),
    "hdrsky_up2x_xf_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_float, c_float, P, P]),
    "hdrsky_act_bwd_bf16": (c_int, [P, P, c_float, c_size_t, P, c_int, P]),
    "hdrsky_concat2": (c_int, [P, c_int, P, c_int, c_size_t, P, P]),
    "hdrsky_debug_wgrad2_stamps": (None, [P]),
    "hdrsky_act_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, c_int, P, P, c_float, c_float, P, P]),
    "hdrsky_concat_rows4": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, c_int, P, P]),
    "hdrsky_vgg_pre": (c_int, [P, c_size_t, P, P]),
    "hdrsky_flip_rgb": (c_int, [P, c_size_t, P, P]),
    "hdrsky_axpby": (c_int, [P, c_float, P, c_float, c_size_t, P, P]),
    "hdrsky_fc_wgrad": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "hdrsky_rmsprop": (c_int, [P, P, P, c_size_t, c_float, c_float, c_float, c_float, P]),
    "hdrsky_rmsprop2": (c_int, [P, P, P, c_size_t, P, P, P, c_size_t, c_float, c_float, c_float, c_float, P]),
    "hdrsky_rmsprop_fc": (c_int, [P, P, P, c_int, c_int, c_float, c_float, c_float, c_float, P, P, P]),
    "hdrsky_fc_wgrad_bf16": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    "hdrsky_fc_xtdy_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "hdrsky_rmsprop_fc_fused": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                        P, P, P, P, P]),
    "hdrsky_rmsprop_fc_fused_bias": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                             P, P, P, P, P, P, P]),
    "hdrsky_rmsprop_fc_fused_prepare": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, P, P, P]),
    "hdrsky_rmsprop_fc_fused_apply": (c_int, [P, P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, P, P]),
    "hdrsky_adam": (c_int, [P, P, P, P, c_size_t, c_float, c_float, c_float, c_float, c_float, P]),
    "hdrsky_resconv_supported": (c_int, [c_int] * 6),
    "hdrsky_resconv": (c_int, [ctypes.POINTER(ResconvArgs), P]),
    "hdrsky_dgb_reduce": (c_int, [P, c_int, c_int, P]),
    "hdrsky_to_bf16": (c_int, [P, P, c_size_t, P]),
}

_lib = None


def load():
    """Loads libhdrsky.so once and types every entry point.  Raises RuntimeError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libhdrsky.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback for the hot path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    # struct-layout contract: a mirror that is smaller than the library's structure would be overrun by the *_init calls
    if lib.hdrsky_abi_version() != ABI_VERSION:
        raise RuntimeError("libhdrsky.so speaks ABI %d, this binding %d - rebuild the library (make -C csrc)" %
                           (lib.hdrsky_abi_version(), ABI_VERSION))
    for cname, mirror in STRUCTS.items():
        if lib.hdrsky_sizeof(cname.encode()) != ctypes.sizeof(mirror):
            raise RuntimeError("%s: the library's layout has %d bytes, the ctypes mirror %d" %
                               (cname, lib.hdrsky_sizeof(cname.encode()), ctypes.sizeof(mirror)))
    _lib = lib
    return lib


class HdrSkyError(RuntimeError):
    pass


_ERR = {-1: "HDRSKY_EINVAL (bad argument / shape)", -2: "HDRSKY_EUNSUPPORTED (configuration not built)",
        -3: "HDRSKY_ELAUNCH (HIP launch error)"}


def check(rc, what):
    if rc != 0:
        raise HdrSkyError("%s failed: %s" % (what, _ERR.get(rc, rc)))
