"""Every HDRSKY_* environment variable the Python host code looks at, read ONCE (at import, or again by `reload()` - the
test-suite changes them inside one process); the C library keeps its own copy (csrc/hooks.h, `hdrsky_hooks_reload`).

SWITCHES select between shipped code paths / deployment options and are always honoured.  TUNING HOOKS belong to the A/B
experiments of profiles/ (stream placement of plan segments, storage-format toggles that were measured and decided); they
are honoured only under HDRSKY_EXPERIMENTS=1 and read as their defaults otherwise.  INTEGRATION.md section 5 lists both."""
import os


class _Hooks:
    def reload(self):
        env = os.environ.get
        self.experiments = env("HDRSKY_EXPERIMENTS", "0") == "1"
        exp = (lambda name, dflt: env(name, dflt)) if self.experiments else (lambda name, dflt: dflt)
        # ---- switches ----------------------------------------------------------------------------------------------
        self.wgrad2 = env("HDRSKY_WGRAD2", "1") != "0"                  # conv_wgrad2_kernel for bf16-operand layers
        self.wgrad_atomic = env("HDRSKY_WGRAD_ATOMIC", "0") == "1"      # fp32-atomics weight gradients (not deterministic)
        self.da_wgrad_region = env("HDRSKY_DA_WGRAD_REGION", "1") != "0"   # LDS-region kernel gradient of the DA conv
        self.dog_fused = env("HDRSKY_DOG_FUSED", "1") != "0"            # the DoG loss term as one launch
        self.da_mat = env("HDRSKY_DA_MAT", "1") != "0"                  # distortion-aware layers: gathered operand written once (bf16) + 1x1 conv
        self.dist_backend = env("HDRSKY_DIST_BACKEND") or None          # "gloo" for the one-card rehearsals
        self.dp_mode = env("HDRSKY_DP_MODE") or None                    # parallel.MODES
        # ---- tuning hooks (HDRSKY_EXPERIMENTS=1) ---------------------------------------------------------------------
        self.da_wgrad_region_maxc = int(exp("HDRSKY_DA_WGRAD_REGION_MAXC", "64"))
        self.inxf_affine_min = int(exp("HDRSKY_INXF_AFFINE_MIN", "64"))
        self.deconv_mat = exp("HDRSKY_DECONV_MAT", "1") != "0"
        self.sun3 = exp("HDRSKY_SUN3", "0") == "1"
        self.vgg_fold = exp("HDRSKY_VGG_FOLD", "1") != "0"
        self.vgg_bf16 = exp("HDRSKY_VGG_BF16", "1") != "0"
        self.nab_dy_bf16 = exp("HDRSKY_NAB_DY_BF16", "1") != "0"
        # raw conv outputs in front of a norm layer as bf16 (ABI 4): measured on the 32x128 step, -0.3 % step time (everything
        # sits in the 256 MB MALL: storage width is not what bounds these launches) for 1.5x the loss-term error - off
        self.raw_bf16 = exp("HDRSKY_RAW_BF16", "0") == "1"
        # the forward conv writes its transformed operand as bf16 for the layer's weight gradient (no hdrsky_act_bf16 launch)
        self.emit_xb = exp("HDRSKY_EMIT_XB", "1") != "0"
        self.disc_split = exp("HDRSKY_DISC_SPLIT", "0") == "1"
        # generator_forward: the encoder branch forks off behind the sun-pose net's conv layers instead of at the input
        self.fwd_stagger = exp("HDRSKY_FWD_STAGGER", "0") == "1"
        # HDRSKY_STREAM_PRIO="p0,p1,p2": priorities of the training step's three streams (torch: -1 = high, 0 = default)
        self.stream_prio = [int(v) for v in exp("HDRSKY_STREAM_PRIO", "0,0,0").split(",")]
        # the perceptual term's prediction pass as two half batches on two streams (1), or as one pass on stream 1 (0)
        self.vgg_split = exp("HDRSKY_VGG_SPLIT", "1") != "0"
        # the sky / sun decoders' layers of equal shape as paired launches on a batch of 2 B (forward heads and the whole backward chain)
        self.dec_pair = exp("HDRSKY_DEC_PAIR", "1") != "0"
        self.dec_head_early = exp("HDRSKY_DEC_HEAD_EARLY", "1") != "0"
        self.bwd_dense_stream = int(exp("HDRSKY_BWD_DENSE_STREAM", "2"))
        self.wg_res_stream = int(exp("HDRSKY_WG_RES_STREAM", "1"))
        # (round 5: with the decoders paired the main chain ends ~300 us before the side streams - the encoder head's weight
        # gradients close it again instead of queueing behind wg_res on stream 1: -0.5 %, profiles/r05_plan_ab_final.txt)
        self.wg_enc_split = exp("HDRSKY_WG_ENC_SPLIT", "0") != "0"
        self.apply_fc_stream = int(exp("HDRSKY_APPLY_FC_STREAM", "2"))
        self.apply_after_fc = exp("HDRSKY_APPLY_AFTER_FC", "0") == "1"
        # HDRSKY_APPLY_FC_CUS="lo,hi[,step]": the Dense update on a fourth stream confined to these compute units (experiment)
        self.apply_fc_cus = [int(v) for v in exp("HDRSKY_APPLY_FC_CUS", "").split(",") if v]
        self.plan_move = exp("HDRSKY_PLAN_MOVE", "")
        self.vgg_target_late = exp("HDRSKY_VGG_TARGET_LATE", "1") != "0"      # the target pass of the perceptual term behind fwd_enc (round 5)
        self.fwd_sun_split = exp("HDRSKY_FWD_SUN_SPLIT", "0") == "1"      # the sun branch as conv + rest segments (scheduling experiments)
        self.plan_merge = exp("HDRSKY_PLAN_MERGE", "auto")  # "a+b+c,d+e": consecutive segments of one stream as ONE segment (one hipGraph); "auto": the trainer's defaults; "": none
        # HDRSKY_FC_FIN=1: Dense layers' bias / activation / mask inside the product's launch (hdrsky_fc_fwd_fin / _dgrad_fin, last-workgroup
        # ticket).  Off: bit-identical but no faster than the hdrsky_fc_finalize launch it saves (step level, forward pass +2 %: the slices
        # must be stored write-through; with agent-scope fences instead, which flush the die's L2, the step was 38 % slower)
        self.fc_fin = exp("HDRSKY_FC_FIN", "0") == "1"
        self.plan_deps = exp("HDRSKY_PLAN_DEPS", "")      # "segment:dependency,...": extra dependencies (scheduling experiments)
        return self


H = _Hooks().reload()


def reload():
    """Re-read the environment here AND in the C library (tests; never while launches are in flight on another thread)."""
    H.reload()
    from . import _lib
    if _lib._lib is not None:
        _lib._lib.hdrsky_hooks_reload()
