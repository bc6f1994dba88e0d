/*
 * hdrsky.h - C ABI of libhdrsky.so, the MI355X (gfx950) compute library behind the
 * LDR->HDR sky-panorama hot path.
 *
 * The reference has no FFI: its boundary is the Keras Layer / Model Python API
 * (SURVEY.md section 8b).  Every entry point below names the reference interface whose
 * arithmetic it replaces (file:line under the reference root).  The Python host layer
 * (the *_amd package) mirrors the reference's class / method names and binds these
 * functions with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes; no C++ / torch types cross the boundary.
 *   - every pointer is DEVICE memory owned by the caller unless marked [host];
 *     the library never allocates, frees or synchronises; all work is enqueued on
 *     the `stream` argument (a hipStream_t passed as void*), so calls are capturable
 *     into a hipGraph.
 *   - tensors are NHWC fp32 at the boundary; `compute` selects how the contractions
 *     run on the matrix cores: HDRSKY_BF16 (one bf16 MFMA product, fp32 accumulate)
 *     or HDRSKY_BF16X3 (hi/lo split operands, three MFMA products: fp32-class accuracy).
 *   - return value: 0 on success, a negative HDRSKY_E* code otherwise; nothing throws,
 *     nothing calls exit().  Functions are re-entrant and thread-safe across streams.
 */
#ifndef HDRSKY_H
#define HDRSKY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDRSKY_OK 0
#define HDRSKY_EINVAL (-1)      /* bad argument / shape */
#define HDRSKY_EUNSUPPORTED (-2) /* configuration not built */
#define HDRSKY_ELAUNCH (-3)     /* HIP launch error */

#define HDRSKY_BF16 0
#define HDRSKY_BF16X3 1

/* input-side fused transform of the conv / fc operand */
#define HDRSKY_IN_NONE 0     /* x as is */
#define HDRSKY_IN_AFFINE 1   /* act(x*scale[b*ss_bstride + c] + shift[...]) */
#define HDRSKY_IN_PARTIALS 2 /* InstanceNorm finalised from the producer's per-tile (sum, sumsq) partials */

const char* hdrsky_version(void);

/* Struct-layout contract of this header.  hdrsky_sizeof("hdrsky_conv_desc" | "hdrsky_wgrad_job" |
 * "hdrsky_resconv_args") = sizeof of that structure as the LIBRARY was compiled (0: unknown name);
 * HDRSKY_ABI_VERSION is bumped whenever a structure or a signature changes (hdrsky_abi_version() returns
 * the library's).  A binding checks both once at load time - the library itself cannot see the caller's
 * layout, and a structure that grew (round 2: hdrsky_conv_desc 25 -> 29 fields) would otherwise be
 * written past the end of a stale mirror by hdrsky_conv_desc_init.  (The reference has no FFI: SURVEY.md
 * section 8b; this is the convention a ctypes / cffi binding needs.) */
#define HDRSKY_ABI_VERSION 4
int hdrsky_abi_version(void);
size_t hdrsky_sizeof(const char* struct_name);

/* Environment switches.  The library reads its HDRSKY_* variables ONCE, at the first call that needs one, into one
 * structure (csrc/hooks.h lists them): switches between shipped code paths (HDRSKY_DA_REGION, HDRSKY_NO_PHASE, ...) are
 * always honoured, the tuning hooks of the A/B experiments (tile shapes, workgroup budgets) only under HDRSKY_EXPERIMENTS=1.
 * hdrsky_hooks_reload [host] reads the environment again (a test-suite that changes variables inside one process); not to
 * be called while another thread is inside a launch function.  hdrsky_experiments_enabled: 1 when the gate was open at
 * the last (re)load.  (The reference reads no environment: its switches are argparse flags, train.py:23-31.) */
int hdrsky_hooks_reload(void);
int hdrsky_experiments_enabled(void);

/* ------------------------------------------------------------------------------------------
 * Convolution family.  Replaces:
 *   ops.conv2d.call            ops.py:41-42     tf.nn.conv2d(x, w, [1,s,s,1], 'SAME') + bias_add
 *   ops.deconv2d.call 'resize' ops.py:121-124   tf.image.resize(BILINEAR) -> stride-1 SAME conv + bias
 *   Keras Conv2D               discriminator.py:11-13, :39-40 ; sunrad_net.py:12-14
 *   vgg16.conv2d.call          vgg16.py:32-36   conv + bias + relu
 * and the fused neighbours tfa InstanceNormalization (generator.py:15,19,61-85,
 * sunpose_net.py:12,17) / Keras BatchNormalization + LeakyReLU (discriminator.py:16-17) that
 * the reference applies between two convolutions: the normalise+activate of the PREVIOUS layer
 * is folded into this conv's operand load, and this conv emits the (sum, sumsq) partials the
 * NEXT normalisation needs.
 * ---------------------------------------------------------------------------------------- */
typedef struct hdrsky_conv_desc {
  int32_t B, H, W, Cin;   /* input tensor x [B,H,W,Cin] */
  int32_t Ho, Wo, Cout;   /* output tensor y [B,Ho,Wo,Cout] */
  int32_t KH, KW, stride; /* filter, stride (1 or 2) */
  int32_t pad_t, pad_l;   /* zero padding before (TF SAME: total//2; VALID: 0) */
  int32_t upsample;       /* 1, or 2: x is bilinearly resized (half-pixel) to [2H,2W] first */
  int32_t dilate;         /* 1, or 2: x is zero-stuffed to [2H-1+dil_extra.., ..] (dgrad of stride 2) */
  int32_t Hc, Wc;         /* conv-input domain after upsample/dilate (host fills: see hdrsky_conv_desc_init) */
  int32_t compute;        /* HDRSKY_BF16 | HDRSKY_BF16X3 */
  /* input transform */
  int32_t in_mode;        /* HDRSKY_IN_* */
  int32_t ss_bstride;     /* AFFINE: Cin for per-(b,c) tables, 0 for per-channel tables */
  int32_t in_nparts;      /* PARTIALS: tiles per sample in in_part */
  float in_eps;           /* PARTIALS: variance epsilon (1e-3 for tfa InstanceNormalization) */
  float in_slope;         /* leaky slope applied after the affine: 1 = none, 0 = relu, 0.1, 0.3 */
  /* epilogue: y = final_relu( act(conv + bias) + residual ) */
  float out_slope;        /* 1 = none, 0 = relu, else leaky slope */
  int32_t final_relu;     /* apply relu after the residual add */
  int32_t want_stats;     /* write per-tile (sum, sumsq) of conv+bias into stats_part */
  /* bf16 activation storage (HDRSKY_BF16 only; both 0 after hdrsky_conv_desc_init): */
  int32_t x_bf16;         /* x points to bf16 [B,H,W,Cin]: a FINAL activation (in_mode NONE, in_slope 1, Cin % 32 == 0,
                             upsample 1) - what the staging would have rounded the fp32 tensor to anyway */
  int32_t y_bf16;         /* y points to bf16 [B,Ho,Wo,Cout] (Cout % 4 == 0): the epilogue result rounded to nearest even;
                             the statistics are still taken from the fp32 values */
  int32_t res_mode;       /* 0: `residual` is an fp32 tensor that is added.  1: `residual` points to a bf16 ACTIVATED tensor of
                             y's shape and the result is multiplied by (it > 0 ? 1 : mask_slope) - the activation backward
                             fused into a data-gradient conv (grad through ReLU: mask_slope 0) */
  float mask_slope;
} hdrsky_conv_desc;

/* Fills Ho/Wo/pad/Hc/Wc for TF padding ("SAME": same=1, "VALID": same=0); returns 0 or HDRSKY_EINVAL. [host] */
int hdrsky_conv_desc_init(hdrsky_conv_desc* d, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                          int stride, int same, int upsample);

/* Descriptor of the DATA GRADIENT of conv `fwd` (stride-1 conv of the zero-stuffed output gradient with the
 * transpose_flip=1 packed filter; output = fwd's conv-input domain). [host] */
int hdrsky_conv_desc_init_dgrad(hdrsky_conv_desc* d, const hdrsky_conv_desc* fwd);

/* Number of bf16 elements of the packed weight image for a [KH,KW,Cin,Cout] filter. [host] */
size_t hdrsky_conv_packed_elems(int KH, int KW, int Cin, int Cout);

/* Packs fp32 HWIO weights w[KH,KW,Cin,Cout] into the MFMA B-operand image (hi plane, and the
 * bf16 residual plane `lo` when non-null).  transpose_flip=1 packs the dgrad filter
 * w'[ky,kx,co,ci] = w[KH-1-ky,KW-1-kx,ci,co] (then Cin/Cout below are those of w'). */
int hdrsky_conv_pack_weights(const float* w, int KH, int KW, int Cin, int Cout, int transpose_flip,
                             void* packed_hi, void* packed_lo, void* stream);

/* Re-packs MANY filters in one launch (after an optimizer step).  jobs: device array [njobs][9] of int64
 * {w ptr, packed_hi ptr, packed_lo ptr (0 = none), KH, KW, Cin, Cout, transpose_flip, first_block}; job j owns blocks
 * [first_block_j, first_block_{j+1}) with ceil(packed_elems/2048) blocks each; total_blocks = their sum. */
int hdrsky_conv_pack_weights_multi(const void* jobs, int njobs, int total_blocks, void* stream);

/* [host] Name of the kernel instantiation hdrsky_conv2d_fwd launches for this descriptor (the tile table of
 * csrc/conv_igemm.hip), as rocprofv3 --kernel-trace prints it: lets bench.py's per-layer roofline rows be matched with
 * the committed kernel statistics under profiles/. */
int hdrsky_conv_kernel_name(const hdrsky_conv_desc* d, char* buf, int n);
/* Number of (sum,sumsq) tiles per sample this descriptor's launch writes to stats_part
 * ([B][nparts][2][Cout] fp32). [host] */
int hdrsky_conv_stats_nparts(const hdrsky_conv_desc* d);

/* y = epilogue(conv(transform(x)))   (see the struct).  Pointers not used by the selected
 * modes may be NULL.  in_scale/in_shift: AFFINE tables; in_part/in_gamma/in_beta: PARTIALS. */
int hdrsky_conv2d_fwd(const hdrsky_conv_desc* d, const float* x, const void* w_hi, const void* w_lo,
                      const float* bias, const float* in_scale, const float* in_shift,
                      const float* in_part, const float* in_gamma, const float* in_beta,
                      const float* residual, float* y, float* stats_part, void* stream);
/* The same launch, ALSO writing xb_out [B,H,W,Cin] bf16 = act(norm(x)): the transformed operand exactly as the matrix cores saw
 * it (every input pixel written once, by the tile that owns it).  It is the x of the layer's weight gradient (a hdrsky_wgrad_job
 * with x_bf16 and no transform: the LDS-DMA kernel), which otherwise needs hdrsky_act_bf16 to produce it from the raw tensor.
 * hdrsky_conv2d_emit_supported: single-product mode, Cin a multiple of 32, stride 1 or 2, no resize / dilation, Cout > 1, and an
 * operand transform to apply. */
int hdrsky_conv2d_emit_supported(const hdrsky_conv_desc* d); /* [host] */
int hdrsky_conv2d_fwd_emit(const hdrsky_conv_desc* d, const float* x, const void* w_hi, const void* w_lo, const float* bias, const float* in_scale, const float* in_shift, const float* in_part, const float* in_gamma, const float* in_beta, const float* residual, float* y, float* stats_part, void* xb_out, void* stream);

/* PAIRED LAUNCHES (round 5).  Two layers of identical geometry whose launches the step used to issue one after the other - the sky and
 * sun decoders of generator.py:110-156, forward and backward - run as ONE launch on a tensor of 2 x the batch: samples [0, B/2) belong to
 * the first layer (its filter, bias, gamma, beta, residual), samples [B/2, B) to the second (the *2 arguments).  Every sample's
 * arithmetic is that of the unpaired launch on its own layer (the conv takes the tile a B/2-sample launch takes), so outputs,
 * statistics partials and everything derived from them are bit-identical to the two separate launches.
 * hdrsky_conv2d_fwd_pair: d->B = 2 x the layers' batch; x_shared != 0: x holds B/2 samples that both layers read (no operand
 * transform then); in_scale / in_shift / in_part are the whole batch's tables as for any launch; no emit / one-channel form. */
int hdrsky_conv2d_fwd_pair(const hdrsky_conv_desc* d, const float* x, int x_shared, const void* w_hi, const void* w_lo, const float* bias,
                           const void* w_hi2, const void* w_lo2, const float* bias2, const float* in_scale, const float* in_shift,
                           const float* in_part, const float* in_gamma, const float* in_beta, const float* in_gamma2,
                           const float* in_beta2, const float* residual, const float* residual2, float* y, float* stats_part,
                           void* stream);
/* hdrsky_in_affine / hdrsky_norm_act_bwd (no atomics form) / hdrsky_up2x_xf_bf16 on a paired tensor: gamma / beta for the first
 * half of the batch, gamma2 / beta2 for the second.  hdrsky_up2x_bwd: bit 2 (value 4) of `accumulate` = dy holds 2 B samples and
 * dx[b] = adjoint(dy[b]) + adjoint(dy[b + B]) (two layers' gradients with respect to an input they share). */
int hdrsky_in_affine_pair(const float* part, int nparts, int B, int C, int count, const float* gamma, const float* beta,
                          const float* gamma2, const float* beta2, float eps, float* scale, float* shift, void* stream);
int hdrsky_norm_act_bwd_pair(const float* x, const float* part, int nparts, const float* gamma, const float* beta, const float* gamma2,
                             const float* beta2, float eps, float slope, const float* dy, int pooled, void* dx, int dx_bf16,
                             float* sums, float* ws, int B, int H, int W, int C, void* stream);
int hdrsky_up2x_xf_bf16_pair(const float* x, int x_bf16, int B, int H, int W, int C, const float* in_part, int in_nparts, const float* gamma,
                             const float* beta, const float* gamma2, const float* beta2, float eps, float slope, void* y_bf16, void* stream);



/* ------------------------------------------------------------------------------------------
 * Normalisation / activation / pooling around the convolutions
 * ---------------------------------------------------------------------------------------- */

/* y = leaky(InstanceNorm(x), slope) [+ residual]; ypool (nullable) = 2x2/2 max-pool of y.
 * IN statistics come from the producer conv's partials part[B][nparts][2][C].
 * Replaces tfa InstanceNormalization + tf.nn.leaky_relu / ops.relu + tf.add + ops.maxpool2d:
 * generator.py:26-35 (resBlock tail), :104-106 ; sunpose_net.py:20-30,55-62 ; ops.py:299-300,328-329. */
int hdrsky_norm_apply(const float* x, int x_bf16, const float* part, int nparts, const float* gamma, const float* beta, float eps,
                      float slope, const float* residual, float* y, float* ypool, int B, int H, int W, int C,
                      void* stream);

/* scale / shift tables [B][C] of the fused operand transform HDRSKY_IN_PARTIALS describes (InstanceNormalization of a conv
 * output from its tile partials, generator.py:26-35), what the consumer kernels derive in that mode (same formula; equal to
 * an ulp or two) - computed once, for HDRSKY_IN_AFFINE consumers (ss_bstride = C): worth it when a sample has many tiles (128x512 maps: 512),
 * because in PARTIALS mode every workgroup of every consumer launch walks all of them. */
int hdrsky_in_affine(const float* part, int nparts, int B, int C, int count, const float* gamma, const float* beta, float eps,
                     float* scale, float* shift, void* stream);
/* mean / rstd / (gamma*rstd) / (beta - mean*gamma*rstd) tables [B][C] from the partials; outputs nullable. */
int hdrsky_in_finalize(const float* part, int nparts, int B, int C, int count, const float* gamma, const float* beta,
                       float eps, float* mean, float* rstd, float* scale, float* shift, void* stream);

/* Keras BatchNormalization in inference mode as a per-channel affine (discriminator.py:25, sunrad_net.py:26
 * with training=False): scale = gamma*rsqrt(moving_var+eps), shift = beta - moving_mean*scale. */
int hdrsky_bn_eval_affine(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                          float eps, int C, float* scale, float* shift, void* stream);

/* Data gradient of y = leaky(IN(x)) [-> 2x2 max-pool when pooled=1]; dy is [B,H,W,C] or [B,H/2,W/2,C].
 * sums (nullable) receives [B][2][C] = (sum g, sum g*xhat) planes; dgamma/dbeta (nullable) are ACCUMULATED with
 * fp32 atomics.  tf.gradients path of grad_cam.py:31 and the IN backward of train.py:402. */
int hdrsky_norm_act_bwd(const float* x, const float* part, int nparts, const float* gamma, const float* beta,
                        float eps, float slope, const float* dy, int pooled, void* dx, int dx_bf16, float* sums,
                        float* dgamma, float* dbeta, float* ws, int B, int H, int W, int C, void* stream);
/* RAW CONV OUTPUTS AS bf16 (round 4, ABI 4).  In the single-product mode a conv output in front of a norm layer may be stored
 * as bf16 (hdrsky_conv_desc.y_bf16 together with want_stats: the statistics still come from the fp32 accumulators).  Its
 * readers widen it on the way in: hdrsky_conv2d_fwd (x_bf16 with an operand transform), the x_bf16 argument of
 * hdrsky_norm_apply / hdrsky_up2x_xf_bf16 / hdrsky_act_bf16, and BIT 2 (value 4) of the dx_bf16 flag word of
 * hdrsky_norm_act_bwd / hdrsky_bn_act_bwd / hdrsky_affine_act_bwd.  Written once, read four to five times per step. */
/* (hdrsky_norm_act_bwd, hdrsky_bn_act_bwd, hdrsky_affine_act_bwd, hdrsky_act_bwd_bf16: bit 1 of dx_bf16 = dy is GIVEN as bf16 - the output of a data-gradient conv with
 * hdrsky_conv_desc.y_bf16 that nothing else reads: half the bytes of the two passes over it.)
 * (dx_bf16 here and in hdrsky_bn_act_bwd / hdrsky_affine_act_bwd / hdrsky_act_bwd_bf16: dx is stored as bf16 - for a gradient
 * whose only readers are a data-gradient conv (hdrsky_conv_desc.x_bf16) and a weight gradient (hdrsky_wgrad_job.dy_bf16),
 * which round it to bf16 while staging anyway: bit-neutral, half the bytes.) */
/* Spatial slices S the call above splits each sample into; when S > 1 it needs the workspace ws [B][S][2][C]. [host] */
int hdrsky_norm_act_bwd_nslices(int B, int H, int W, int C, int pooled);
/* 1 when hdrsky_norm_act_bwd runs a call that stores dx as bf16 (bit 0 of dx_bf16: the single-product mode's calls) as ONE launch
 * that reads x and dy once - the (sample, 8- or 16-channel group) slab held in the registers of one workgroup: 256 / 1024 pixels
 * (256 / 1024 windows of the pooled form) per sample; the workspace is then not touched.  HDRSKY_NAB_ONE=0 switches it
 * off. [host] */
int hdrsky_norm_act_bwd_one_launch(int H, int W, int pooled, int dy_bf16);

/* ------------------------------------------------------------------------------------------
 * Sun-pose dense head (sunpose_net.py:48-52,64-70) and its Grad-CAM backward (grad_cam.py:29-44)
 * ---------------------------------------------------------------------------------------- */
int hdrsky_fc_pack_weights(const float* w, int K, int N, void* packed_hi, void* packed_lo, void* natural_hi,
                           void* natural_lo, void* stream);
int hdrsky_fc_nsplit(int R); /* [host] reduction split the fc kernels use for a reduction length R */
int hdrsky_fc_fwd(const float* x, const void* packed_hi, const void* packed_lo, int M, int K, int N, int nsplit,
                  int compute, float* out_part, void* stream);
int hdrsky_fc_dgrad(const float* dy, const void* natural_hi, const void* natural_lo, int M, int K, int N, int nsplit,
                    int compute, float* dx_part, void* stream);
/* hdrsky_fc_fwd / hdrsky_fc_dgrad + hdrsky_fc_finalize in ONE launch (round 5): the workgroup that finishes last among the nsplit
 * reduction slices of a 64-column block adds the slices in slice order - bit-identical to the two launches.  part_ws [nsplit][M][N or K]:
 * the partial sums (workspace); tickets: ceil(columns / 64) zero-initialised 32-bit words owned by the call site (self-resetting;
 * launches that share them must not overlap); bias / mask_src / zero_word as hdrsky_fc_finalize (sunpose_net.py:48-51, grad_cam.py:31). */
int hdrsky_fc_fwd_fin(const float* x, const void* packed_hi, const void* packed_lo, int M, int K, int N, int nsplit, int compute, float* part_ws, void* tickets, const float* bias, int relu, const float* mask_src, float* y, void* zero_word, void* stream);
int hdrsky_fc_dgrad_fin(const float* dy, const void* natural_hi, const void* natural_lo, int M, int K, int N, int nsplit, int compute, float* part_ws, void* tickets, const float* bias, int relu, const float* mask_src, float* dx, void* zero_word, void* stream);
/* y = [relu](sum_s part[s] + bias) [* (mask_src > 0)]; zero_word (nullable): a 4-byte word this launch clears - the max
 * accumulator of a hdrsky_softmax_head later on the same stream, saving a memset launch in the chain. */
int hdrsky_fc_finalize(const float* part, int nsplit, int M, int N, const float* bias, int relu, const float* mask_src,
                       float* y, void* zero_word, void* stream);
/* z = relu(sum_s part[s] + bias); cmf = softmax(z); *gmax_bits = max(*gmax_bits, bits(max cmf)) (zero it first). */
int hdrsky_softmax_head(const float* part, int nsplit, int M, int N, const float* bias, float* z, float* cmf,
                        void* gmax_bits, void* stream);
/* *gmax_bits = max(*gmax_bits, bits(max_i x[i])), x >= 0 (zero the word first): tf.reduce_max(sunpose_pred)
 * (generator.py:160) for a sun-position map that is an INPUT of the step (the 128x512 configuration, SURVEY.md
 * section 8d: its 12.9 G-parameter sun-pose net is substituted by its outputs). */
int hdrsky_global_max(const float* x, size_t n, void* gmax_bits, void* stream);
/* hdrsky_softmax_head and hdrsky_softmax_pick_bwd in one launch (the row is in registers anyway): additionally
 * dz = d cmf[m, c] / d z with c = first argmax of pick_src[m, :] - or of cmf[m, :] itself when pick_src is NULL
 * (inference.py:98) - and idx_out[m] = c (nullable).  Same arithmetic and tie rule as the two separate launches. */
int hdrsky_softmax_head_pick(const float* part, int nsplit, int M, int N, const float* bias, float* z, float* cmf,
                             void* gmax_bits, const float* pick_src, float* dz, int* idx_out, void* stream);
/* dz = d cmf[m, argmax(pick_src[m])] / d z  (inference.py:98 ; train.py:265-267) */
int hdrsky_softmax_pick_bwd(const float* cmf, const float* z, const float* pick_src, int M, int N, float* dz,
                            int* idx_out, void* stream);
/* out[b][c] = scale * sum_p x[b][p][c] */
int hdrsky_spatial_sum(const float* x, int B, int P, int C, float scale, float* out, void* stream);
/* cam[b][p] = relu(sum_c w[b][c]*A[b][p][c])  (grad_cam.py:34-38).  w_nparts == 0: w is a [B][C] table;
 * w_nparts > 0: w is the statistics tensor [B][w_nparts][2][C] of the conv that produced the activation
 * gradient and its per-tile sums are reduced here; w_nparts < 0: w is the activation gradient itself, [B][-w_nparts
 * pixels][C] (at most 256 pixels: the sum is repeated by every block of a sample); the weights are multiplied by w_scale. */
int hdrsky_grad_cam(const float* A, const float* w, int w_nparts, float w_scale, int B, int P, int C, float* cam,
                    void* stream);
/* the three maps of one sweep (grad_cam.py:52-60: layer1, layer2, layer3) in one launch; every argument is a host array of
 * three with the meaning it has in hdrsky_grad_cam; the values are those of three separate calls */
int hdrsky_grad_cam3(const float* const* A, const float* const* w, const int* w_nparts, const float* w_scale, const int* P,
                     const int* C, float* const* cam, int B, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sun-radiance head, tone mapping, blending
 * ---------------------------------------------------------------------------------------- */
/* plz = concat(ldr, cam1, resize(cam2), resize(cam3))  (generator.py:161-164) -> [B,H,W,6] */
int hdrsky_plz_build(const float* ldr, const float* cam1, const float* cam2, const float* cam3, int B, int H, int W,
                     float* plz, void* stream);
/* sunRadNet heads, stage 1 (sunrad_net.py:52-53): part[b][s][2] = slice s of flatten(leaky(x*scale[c]+shift[c])) . {kg, kb};
 * F = flatten length, S slices per sample. */
int hdrsky_dense_heads(const float* x, const float* scale, const float* shift, float slope, int B, int F, int C,
                       const float* kg, const float* kb, int S, float* part, void* stream);
/* stage 2 + Dirac-delta radiance: gamma/beta = sigmoid(sum_s part + bias) (sunrad_net.py:54-59), then the radiance map
 * (sunrad_net.py:61-69, generator.py:160,167) and its hdr_logCompression (tf_utils.py:263-271), tiled to 3 channels. */
int hdrsky_sun_rad(const float* cmf, const void* gmax_bits, const float* part, int S, const float* bg, const float* bb,
                   int B, int P, float* gamma_out, float* beta_out, float* rad_lin3, float* rad_gamma3, void* stream);
/* alpha mask + blend + log decompression (inference.py:91-94,109-113; train.py:258-261,293-299); outputs after
 * y_lin are nullable */
int hdrsky_blend(const float* sky_gamma, const float* sun_gamma, int npix, float thr, float* y_gamma, float* y_lin,
                 float* alpha, float* sky_lin, float* sun_lin, void* stream);
/* tf_utils.hdr_logCompression (decompress=0) / hdr_logDecompression (1)  (tf_utils.py:263-280) */
int hdrsky_tonemap(const float* x, float* y, size_t n, int decompress, void* stream);
/* ops.relu (ops.py:324-329; slope 0) and stand-alone LeakyReLU layers: y = x > 0 ? x : slope * x. */
int hdrsky_leaky_relu(const float* x, float* y, size_t n, float slope, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-step kernels (train.py:382-415)
 * ---------------------------------------------------------------------------------------- */

/* Weight / bias gradient of the convolution described by `d` (same descriptor, same fused operand transform as the
 * forward call): dw[KH,KW,Cin,Cout] += sum_pixels X' (x) dY, db[Cout] += sum_pixels dY (db nullable).
 * fp32 atomics: ZERO dw / db first.  tf.GradientTape through tf.nn.conv2d (train.py:402-406). */
int hdrsky_conv2d_wgrad(const hdrsky_conv_desc* d, const float* x, const float* dy, const float* in_scale,
                        const float* in_shift, const float* in_part, const float* in_gamma, const float* in_beta,
                        float* dw, float* db, void* stream);

/* Several independent weight gradients in as few launches as possible (layers of similar geometry share a launch;
 * a training step has ~40 of them and each is far too small for 256 CUs on its own).  `jobs` is a HOST array; the
 * device pointers in it follow hdrsky_conv2d_wgrad.  Same semantics as njobs single calls on `stream`. */
typedef struct hdrsky_wgrad_job {
  hdrsky_conv_desc desc;
  const float* x;
  const float* dy;
  const float* in_scale;
  const float* in_shift;
  const float* in_part;
  const float* in_gamma;
  const float* in_beta;
  float* dw;
  float* db;
  int32_t x_bf16;   /* 1: x points to bf16 data (final activations of the bf16 chain, see hdrsky_resconv; needs */
  int32_t dy_bf16;  /* 1: dy points to bf16 data                             in_mode NONE / Cin, Cout % 8 == 0) */
  /* Distortion-aware layer (distortion_aware_ops.py:5-270, kernel [k*k*C, F]): da_ksize = k > 0 makes the job the weight
   * gradient dW[k*k*C][F] = G^T dY without G in memory - desc = the 1x1 layer over Cin = k*k*da_C virtual channels at the
   * map size, x = the layer's input [B,H,W,da_C] fp32 (da_C % 32 == 0, no operand transform), da_offs = hdrsky_da_offsets'
   * table [H][k*k][2]; the bilinear samples (:62-113) are recomputed while the operand tile is staged. */
  const float* da_offs;
  int32_t da_ksize;
  int32_t da_C;
} hdrsky_wgrad_job;
int hdrsky_conv2d_wgrad_multi(const hdrsky_wgrad_job* jobs, int njobs, void* stream);
/* The same weight gradients, BIT-REPRODUCIBLE: hdrsky_conv2d_wgrad(_multi) splits the pixel reduction over workgroups
 * and adds their partial blocks into dw with fp32 atomics (arrival order = summation order).  Here every workgroup
 * stores its partial block into `ws` (device scratch, hdrsky_conv2d_wgrad_ws_bytes(jobs, njobs) bytes, 16-byte aligned)
 * and a second launch per group adds the partials to dw / db in a fixed order.  [host] for the size query. */
size_t hdrsky_conv2d_wgrad_ws_bytes(const hdrsky_wgrad_job* jobs, int njobs);
int hdrsky_conv2d_wgrad_multi_det(const hdrsky_wgrad_job* jobs, int njobs, void* ws, size_t ws_bytes, void* stream);
/* [host] The kernels one hdrsky_conv2d_wgrad_multi_det call on these jobs launches, " + "-separated, named as rocprofv3
 * prints them (e.g. "conv_wgrad2_kernel<5> + wgrad_reduce_kernel"): bench.py labels its roofline rows with it. */
int hdrsky_conv2d_wgrad_kernel_names(const hdrsky_wgrad_job* jobs, int njobs, char* buf, int n);
/* [host] 1 when the LDS-DMA weight-gradient kernel (conv_wgrad2_kernel: both operands final bf16 tensors, copied into LDS
 * without touching a register) takes this job; with_final_bf16_x != 0 asks whether it WOULD once x is replaced by the
 * final bf16 tensor hdrsky_act_bf16 writes - the host's question in front of that launch (kernels._materialise_bf16_operand). */
int hdrsky_wgrad2_eligible(const hdrsky_wgrad_job* job, int with_final_bf16_x);

/* Keras BatchNormalization(training=True) statistics from the producing conv's partials [nparts_total][2][C]
 * (discriminator.py:25, sunrad_net.py:26): mean/rstd/scale/shift tables + moving-stat update (momentum 0.99,
 * Bessel-corrected variance).  moving_* nullable. */
int hdrsky_bn_train_finalize(const float* part, int nparts_total, int C, int count, const float* gamma, const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, float* mean, float* rstd, float* scale, float* shift, int rows, void* stream);
/* (scale / shift are written as `rows` identical rows of a [rows][C] table: rows = 1 for per-channel tables, rows = B to
 * fill the per-sample affine tables of a batch that carries several BatchNorm groups side by side.)
 * Zero-fill of device memory on `stream` (a kernel launch, deliberately not hipMemsetAsync: memset nodes of a captured
 * hipGraph wrote wrong patterns from the second replay on, profiles/repro_memset_node.py): gradient / loss accumulators. */
int hdrsky_zero(void* p, size_t nbytes, void* stream);
/* [host] number of reduction blocks hdrsky_bn_act_bwd uses; its workspace is (2*nblocks*C + 2*C) floats. */
int hdrsky_bn_bwd_nblocks(void);
/* Backward of y = leaky(BN_train(x)): dx, and dgamma/dbeta ACCUMULATED into the given buffers (nullable). */
int hdrsky_bn_act_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, const float* beta, float slope, int npix, int C, float* workspace, float* dgamma, float* dbeta, void* dx, int dx_bf16, void* stream);
/* The same in two halves, for a BatchNorm whose batch is spread over several data-parallel replicas (the optional
 * "one batch-256 step" semantics of SURVEY.md section 8e; Keras BatchNormalization over the GLOBAL batch,
 * discriminator.py:16,24, sunrad_net.py:17,25).  _reduce writes this replica's hdrsky_bn_bwd_nblocks() partial blocks
 * part [nblocks][2][C] = per-block (sum g, sum g*xhat); the caller all-gathers them; _apply takes the gathered blocks
 * (part_all, nblocks_all, count_all = pixels of the global batch) for the two means of the formula and this replica's
 * own blocks (part_local) for d gamma / d beta (+=; the gradient exchange then sums them over the replicas).
 * m1m2: 2*C floats of scratch. */
int hdrsky_bn_act_bwd_reduce(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, float slope, int npix, int C, float* part, void* stream);
int hdrsky_bn_act_bwd_apply(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, float slope, int npix, int C, const float* part_all, int nblocks_all,
                            double count_all, const float* part_local, int nblocks_local, float* m1m2, float* dgamma,
                            float* dbeta, void* dx, int dx_bf16, void* stream);
/* dx = dy*act'(x*scale[c]+shift[c])*scale[c] (BN in inference mode); scale==NULL: x is the ACTIVATED tensor and
 * dx = dy*(x>0 ? 1 : slope)  (Keras LeakyReLU / ReLU backward). */
int hdrsky_affine_act_bwd(const float* x, const float* dy, const float* scale, const float* shift, float slope, size_t n, int C, void* dx, int dx_bf16, void* stream);
/* 2x2/2 max-pool (vgg16.py:85-86). */
int hdrsky_maxpool_fwd(const float* y, int B, int H, int W, int C, float* p, void* stream);
/* Backward of maxpool(relu(.)) given the post-ReLU tensor y: gradient to the first arg-max where y > 0. */
int hdrsky_maxpool_relu_bwd(const float* y, const float* dp, int B, int H, int W, int C, float* dy, void* stream);
/* y = resize2x_bilinear(a - b) (b nullable), half-pixel centres (tf_utils.py:64). */
int hdrsky_up2x_fwd(const float* a, const float* b, int B, int H, int W, int C, float* y, void* stream);
/* Exact adjoint of the 2x bilinear resize: dx (+)= scale * R^T dy   (backward of ops.py:122 / tf_utils.py:64).
 * accumulate: bit 0 = add to dx, bit 1 = dy is given as bf16 (a data-gradient conv's hdrsky_conv_desc.y_bf16 output). */
int hdrsky_up2x_bwd(const float* dy, int B, int H, int W, int C, float scale, int accumulate, float* dx, void* stream);
/* TFA gaussian_filter2d 3x3, REFLECT padding (tf_utils.py:65,69-70); transpose=1 applies the adjoint. */
int hdrsky_blur3(const float* x, int B, int H, int W, int C, float sigma, int transpose, float* y, void* stream);
/* DoG differences + L1 (tf_utils.py:66-71, train.py:319-322) on the blurred base image: loss += weight*sum_i mean|d_i|,
 * h[5][n] = gradient wrt the five Gaussian images. */
int hdrsky_dog_mid(const float* base, int B, int H, int W, int C, float weight, float* h, float* loss, void* stream);
/* dbase = sum_j G(s_j)^T h_j. */
int hdrsky_dog_mid_bwd(const float* h, int B, int H, int W, int C, float* dbase, void* stream);
/* The whole DoG term of train.py:316-322 (tf_utils.py:61-73) in one launch: *loss += sum_i mean|DoG_i(y) - DoG_i(t)|,
 * dy += weight * its gradient wrt y ([B,H,W,C] fp32; the five calls above chained through LDS bands - rows longer than
 * 1 024 floats at 2x resolution in column strips -, same operators).  HDRSKY_EUNSUPPORTED for more than 64 channels. */
int hdrsky_dog_loss(const float* y, const float* t, int B, int H, int W, int C, float weight, float* loss, float* dy, void* stream);
/* loss += wl*mean|a-b| (b nullable); da (+)= wg*sign(a-b)/n  (train.py:311-313,325). */
int hdrsky_l1(const float* a, const float* b, size_t n, float wl, float wg, float* loss, float* da, int accumulate, void* stream);
/* LSGAN terms (train.py:234-237): loss += wl*mean((x-target)^2); dx = wg*2(x-target)/n. */
int hdrsky_mse(const float* x, float target, size_t n, float wl, float wg, float* loss, float* dx, void* stream);
/* Keras KLDivergence (train.py:232,305) and its gradient wrt cmf. */
int hdrsky_kl(const float* gt, const float* cmf, int B, int N, float* loss, float* dcmf, void* stream);
/* dz = cmf*(dcmf - <dcmf,cmf>)*[z>0]  (softmax + ReLU, sunpose_net.py:68-70). */
int hdrsky_softmax_bwd(const float* cmf, const float* dcmf, const float* z, int M, int N, float* dz, void* stream);
/* Backward of hdrsky_blend (alpha is a constant, train.py:257): dyg / dyl nullable. */
int hdrsky_blend_bwd(const float* y_gamma, const float* alpha, const float* dyg, const float* dyl, size_t n, float* dsky, float* dsun, void* stream);
/* hdrsky_blend_bwd + both hdrsky_decoder_tail_bwd in one launch, with the adversarial term's gradient (channels 3..5 of the
 * discriminator's 6-channel input gradient din6 [pixels][6], NULL allowed) added to dyl on the way: the first stretch of the
 * generator's backward pass (three-channel tensors of n elements; dres_u = the gradient wrt the sun decoder's residual input). */
int hdrsky_head_bwd(const float* y_gamma, const float* alpha, const float* dyg, const float* dyl, const float* din6, const float* y_f, const float* res_f, const float* y_u, const float* res_u, size_t n, float* dc_f, float* dc_u, float* dres_u, void* stream);
/* Backward of y = relu(res + lrelu(c,0.1)) (generator.py:119-124,151-155): dc and (nullable) dres. */
int hdrsky_decoder_tail_bwd(const float* y, const float* res, const float* dy, size_t n, float* dc, float* dres, void* stream);
/* Backward of hdrsky_sun_rad: dpre[B][2] (pre-sigmoid gamma/beta), dcmf += (incl. the batch reduce_max term,
 * generator.py:160: elements tied for the maximum share its gradient evenly, as tf.reduce_max's gradient does - counted
 * with integer atomics, so the result does not depend on the order of arrival).  scratch: B*P + B floats + 1 int. */
int hdrsky_sun_rad_bwd(const float* cmf, const void* gmax_bits, const float* gamma, const float* beta, const float* drg3, int B, int P, float* scratch, float* dpre, float* dcmf, void* stream);
/* The same in two halves, for tf.reduce_max(sunpose_pred) (generator.py:160) taken over the batch of EVERY data-parallel
 * replica: _reduce leaves d(cmf/max) in scratch [B*P] and, behind it, the record (dotx[B], tie count as int bits) =
 * scratch + B*P, B+1 words; the caller all-gathers the records; _apply adds d cmf with the maximum's gradient term summed
 * over the nrec records (rec = the local record and nrec = 1 reproduce hdrsky_sun_rad_bwd). */
int hdrsky_sun_rad_bwd_slices(int P); /* [host] pixel slices per sample; scratch of hdrsky_sun_rad_bwd* = B*P + B + 4 + 3*B*slices floats */
int hdrsky_sun_rad_bwd_reduce(const float* cmf, const void* gmax_bits, const float* gamma, const float* beta, const float* drg3,
                              int B, int P, float* scratch, float* dpre, void* stream);
int hdrsky_sun_rad_bwd_apply(const float* cmf, const void* gmax_bits, const float* scratch, const float* rec, int nrec, int B,
                             int P, float* dcmf, void* stream);
/* Backward of the two Dense(1) heads: dact (wrt the activated flatten), dkg/dkb/dbg/dbb accumulated. */
int hdrsky_dense_heads_bwd(const float* x, const float* scale, const float* shift, float slope, int B, int F, int C, const float* kg, const float* kb, const float* dpre, float* dact, float* dkg, float* dkb, float* dbg, float* dbb, void* stream);
/* out (+)= scale * x[..., c_off:c_off+c_take]  (gradient of tf.concat, discriminator.py:43). */
int hdrsky_slice_channels(const float* x, size_t npix, int C, int c_off, int c_take, float scale, int accumulate, float* out, void* stream);
/* out[npix][Cpad] = x[npix][C] followed by Cpad - C zero channels (the operand a distortion_aware_ops.conv2d with fewer
 * than 32 input channels - sunpose_net.py:11 on the RGB image - is run on; also pads a filter's input-channel axis). */
int hdrsky_pad_channels(const float* x, size_t npix, int C, int Cpad, float* out, void* stream);
/* The ReLU-only VGG16 chain on bf16 activations (vgg16.py:88-165; bit-neutral for the next conv, which rounds its
 * operand to bf16 anyway): 2x2 max-pool of a bf16 map -> fp32 pool (the perceptual feature) and / or bf16 pool (the next
 * conv's operand); the pool + ReLU backward and the plain activation backward with the ACTIVATED tensor given as bf16. */
int hdrsky_maxpool_fwd_bf16(const void* y_bf16, int B, int H, int W, int C, float* p_f32, void* p_bf16, void* stream);
int hdrsky_maxpool_relu_bwd_bf16(const void* y_bf16, const float* dp, int B, int H, int W, int C, void* dy, int dy_bf16,
                                 void* stream);   /* dy_bf16: dy is stored as bf16 (the operand of the next data-gradient conv) */
/* ... with the L1 term of the block's pooled features folded in (the perceptual term, train.py:308-313: pool = pool_i(vgg(pred)),
 * target = pool_i(vgg(hdr_t)), both fp32 [B,H/2,W/2,C]): dp' = (dp or 0, NULL allowed) + wg * sign(pool - target) / n is what gets
 * routed, *loss += wl * mean|pool - target| - the hdrsky_l1 launch in front of every pool backward of the VGG16 pass. */
int hdrsky_maxpool_relu_l1_bwd_bf16(const void* y_bf16, const float* pool, const float* target, const float* dp, int B, int H, int W, int C, float wl, float wg, float* loss, void* dy, int dy_bf16, void* stream);
int hdrsky_act_bwd_bf16(const void* y_bf16, const float* dy, float slope, size_t n, void* dx, int dx_bf16, void* stream);
/* The operand of a resize-deconvolution (ops.py:44-126 method 'resize': tf.image.resize 2x, then the conv) as a bf16
 * tensor: y [B,2H,2W,C] = bf16(resize2x(leaky(IN(x), slope))) with the InstanceNorm affine from the producing conv's
 * statistics partials in_part [B][in_nparts][2][C] (NULL: x is already an activation) - the arithmetic and operation
 * order of hdrsky_conv2d_fwd's upsample = 2 staging, so a plain conv (x_bf16) on y equals the fused one.  C % 8 == 0. */
int hdrsky_up2x_xf_bf16(const float* x, int x_bf16, int B, int H, int W, int C, const float* in_part, int in_nparts, const float* gamma,
                        const float* beta, float eps, float slope, void* y_bf16, void* stream);
/* The activated input of a conv as a final bf16 tensor: y [B,HW,C] = bf16(leaky(x * scale + shift, slope)) with the
 * operand transform of hdrsky_conv_desc (in_mode / in_slope / tables: InstanceNorm + tf.nn.leaky_relu, generator.py:15-19,
 * 61-85, sunpose_net.py:12-18; BatchNormalization + LeakyReLU, discriminator.py:16-17) - the arithmetic of the conv /
 * weight-gradient staging.  Operand of the LDS-DMA weight-gradient path (hdrsky_wgrad_job.x_bf16).  C % 8 == 0, C <= 1024. */
int hdrsky_act_bf16(const float* x, int x_bf16, int B, int HW, int C, int in_mode, const float* in_scale, const float* in_shift, int ss_bstride,
                    const float* in_part, int in_nparts, const float* gamma, const float* beta, float eps, float slope,
                    void* y_bf16, void* stream);
/* debug only: 8 x u64 of s_memtime phase stamps per workgroup of subsequent conv_wgrad2_kernel launches (null disables):
 * [start, ring primed, main loop done, cycles waiting for copies, issuing copies, computing, tiles, end] */
void hdrsky_debug_wgrad2_stamps(void* buf);
/* tf.concat([a, b], axis=-1) (discriminator.py:43). */
int hdrsky_concat2(const float* a, int Ca, const float* b, int Cb, size_t npix, float* out, void* stream);
/* Row-wise concatenation of four [M, w_i] fp32 matrices into out [M, w0+w1+w2+w3] (w_i % 4 == 0, 16-byte aligned
 * bases): the operand block (flat | df1 | f1 | dz) a data-parallel replica contributes to the all-gather from which
 * every replica recomputes the Keras Dense kernels' tape.gradient on the global batch (sunpose_net.py:48-51,
 * train.py:402; <pkg>/parallel.py mode gather_dense). */
int hdrsky_concat_rows4(const float* s0, int w0, const float* s1, int w1, const float* s2, int w2, const float* s3, int w3,
                        int M, float* out, void* stream);
/* x*255 - VGG_MEAN (vgg16.py:133-141). */
int hdrsky_vgg_pre(const float* x, size_t n, float* y, void* stream);
/* tf_utils.rgb2bgr / bgr2rgb (tf_utils.py:85-93): channel reversal of npix 3-channel pixels; x == y is allowed. */
int hdrsky_flip_rgb(const float* x, size_t npix, float* y, void* stream);
/* y = sa*a + sb*b (b nullable). */
int hdrsky_axpby(const float* a, float sa, const float* b, float sb, size_t n, float* y, void* stream);
/* Dense weight / bias gradient dW[K][N] (+)= x^T dy, db (+)= sum_m dy (M <= 32). */
int hdrsky_fc_wgrad(const float* x, const float* dy, int M, int K, int N, int accumulate, float* dw, float* db, void* stream);
/* Keras-2 OptimizerV2 RMSprop step over one flat buffer (train.py:201-202,403,406): g is first multiplied by gscale. */
int hdrsky_rmsprop(float* w, const float* g, float* ms, size_t n, float lr, float rho, float eps, float gscale, void* stream);
/* the same step over two flat buffers (two optimizers' parameters) in one launch */
int hdrsky_rmsprop2(float* w1, const float* g1, float* ms1, size_t n1, float* w2, const float* g2, float* ms2, size_t n2, float lr, float rho, float eps, float gscale, void* stream);
/* The same update for a Dense kernel w[K][N], fused with the refresh of its bf16 images (hdrsky_fc_pack_weights layouts:
 * packed_hi [K/8][N][8], natural_hi [K][N] or NULL) - HDRSKY_BF16 mode only (no residual planes). */
int hdrsky_rmsprop_fc(float* w, const float* g, float* ms, int K, int N, float lr, float rho, float eps, float gscale,
                      void* packed_hi, void* natural_hi, void* stream);
/* The Dense weight gradient on the matrix cores: dW[K][N] (+)= x^T dy with both operands rounded to bf16 (fp32
 * accumulation; HDRSKY_BF16's contract), db (+)= fp32 column sums of dy.  x [M][ldx >= K], dy [M][ldy >= N] (row strides
 * in floats, multiples of 4, 16-byte aligned bases), any M >= 1; K a multiple of 32, N of 256.  Replaces tf.gradients through
 * Keras Dense (sunpose_net.py:48-51,65-68) for the materialised gradient (all-reduce, inspection). */
int hdrsky_fc_wgrad_bf16(const float* x, int ldx, const float* dy, int ldy, int M, int K, int N, int accumulate, float* dw,
                         float* db, void* ws, void* stream);
/* Scratch both entry points need (the bf16 transposed operand images their first launch writes): ws of this many bytes. */
size_t hdrsky_fc_xtdy_ws_bytes(int M, int K, int N);
/* hdrsky_rmsprop_fc with the gradient g = gscale * x^T dy recomputed tile by tile inside the update (operands as for
 * hdrsky_fc_wgrad_bf16) instead of read from memory: w, ms and the bf16 images are updated in place, the weight gradient is
 * never written.  db (nullable) receives the bias gradient (unscaled) for a following hdrsky_rmsprop of the bias. */
int hdrsky_rmsprop_fc_fused(float* w, float* ms, const float* x, int ldx, const float* dy, int ldy, int M, int K, int N,
                            float lr, float rho, float eps, float gscale, void* packed_hi, void* natural_hi, float* db,
                            void* ws, void* stream);
/* ... that also applies the RMSprop step to the layer's bias vector (bias, bias_ms [N]; db [N] - the column sums of dy - must be
 * given), inside the launch that forms those sums. */
int hdrsky_rmsprop_fc_fused_bias(float* w, float* ms, const float* x, int ldx, const float* dy, int ldy, int M, int K, int N, float lr, float rho, float eps, float gscale, void* packed_hi, void* natural_hi, float* db, float* bias, float* bias_ms, void* ws, void* stream);
/* hdrsky_rmsprop_fc_fused_bias as two calls with the same launches, so that the update can be issued later than its operands exist
 * (the next step's forward pass has an idle stream; the end of the step has none): _prepare writes the operands' transposed bf16
 * images into ws (hdrsky_fc_xtdy_ws_bytes), db and - given bias / bias_ms - the bias vector's step; _apply contracts ws and updates
 * w, ms and the two bf16 images.  Between the two calls ws must stay untouched; x and dy may be rewritten. */
int hdrsky_rmsprop_fc_fused_prepare(const float* x, int ldx, const float* dy, int ldy, int M, int K, int N, float lr, float rho, float eps,
                                    float gscale, float* db, float* bias, float* bias_ms, void* ws, void* stream);
int hdrsky_rmsprop_fc_fused_apply(float* w, float* ms, int M, int K, int N, float lr, float rho, float eps, float gscale, void* packed_hi,
                                  void* natural_hi, const void* ws, void* stream);

/* tf.keras.optimizers.Adam (train_sun.py:191 / tf_utils.py:324; defaults beta 0.9 / 0.999, eps 1e-7) over a flat buffer:
 * m, v are the slots; lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) is computed by the caller for step t; g is scaled by gscale. */
int hdrsky_adam(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float beta1, float beta2, float eps,
                float gscale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Device-side input synthesis (train.py:42-94: the host augmentation)
 * ---------------------------------------------------------------------------------------- */
/* `_preprocessing` (train.py:54-94): hdr_t = relu(hdr*t + n_s*(sigma_s*hdr*t) + sigma_c*n_c) per sample b / channel c;
 * ldr = round_half_even(255 * CRF_b(clip(hdr_t, 0, 1))) / 255 with CRF_b a crf_len-sample LUT interpolated linearly
 * (tf_utils.apply_rf, tf_utils.py:245-255).  t [B], sigma_* [B,3], noise_* [B,H,W,3] standard normals, crf [B,crf_len]. */
int hdrsky_ldr_synth(const float* hdr, const float* t, const float* sigma_s, const float* sigma_c, const float* noise_s,
                     const float* noise_c, const float* crf, int crf_len, int B, int H, int W, float* hdr_t, float* ldr,
                     void* stream);
/* The JPEG step of `_preprocessing` (train.py:86-92: `tf.image.adjust_jpeg_quality(img_u8, q)` per sample =
 * libjpeg baseline encode - 4:2:0, slow-integer DCT, Annex-K tables scaled by quality q, force_baseline - and decode -
 * slow-integer IDCT, fancy chroma upsampling; the lossless entropy coding is skipped).  Integer arithmetic, bit-exact
 * against libjpeg.  ldr / out [B,H,W,3] float holding k/255 (channel 0 = R, or B when bgr != 0); quality [B] int32 on the
 * device (train.py:89: round(i/(B-1)*10+90)); ws: hdrsky_jpeg_roundtrip_ws_bytes(B,H,W) bytes of device scratch, 8-byte
 * aligned; ldr / out 16-byte aligned.  Any H, W: partial 16x16 MCUs are completed by libjpeg's edge replication rules.
 * out may alias ldr. */
size_t hdrsky_jpeg_roundtrip_ws_bytes(int B, int H, int W);
int hdrsky_jpeg_roundtrip(const float* ldr, const int* quality, int B, int H, int W, int bgr, unsigned char* ws, float* out,
                          void* stream);
/* `vMF` (train.py:42-52): out[b][j] = exp(kappa*<bin_j, sun(azimuth, elevation[b])>) normalised over the H*W sky bins. */
int hdrsky_vmf_target(const float* elevation, float azimuth, int B, int H, int W, float kappa, float* out, void* stream);

/* [host] CRC-32C (Castagnoli) of n HOST bytes continuing from `crc` (0 to start): the checksum of the TFRecord framing
 * (TensorBoard event files, tf_utils.py:282-292) and of the TF tensor-bundle checkpoint format (train.py:208-220). */
unsigned int hdrsky_crc32c(const void* data, size_t n, unsigned int crc);

/* ------------------------------------------------------------------------------------------
 * Distortion-aware panoramic convolution (distortion_aware_ops.py)
 * ---------------------------------------------------------------------------------------- */
/* conv2d.distortion (distortion_aware_ops.py:198-270), float32 in the reference's operation order:
 * out[h][k*k][2] = (y, x) sampling offsets per row (identical for every column). [host] */
int hdrsky_da_offsets(int h, int w, int ksize, int dilation_rate, int skydome, float* out);
/* conv2d.call (distortion_aware_ops.py:50-123): bilinear gather with y clamp / 360-degree x wrap fused into an MFMA
 * GEMM; w_* = hdrsky_conv_pack_weights image of the [k*k*Cin, Cout] kernel viewed as [k,k,Cin,Cout]; offs = device
 * copy of hdrsky_da_offsets(H, W, k, ..).  stride 1, Cin % 32 == 0.  deconv2d.call (:321-395) = hdrsky_up2x_fwd + this.
 * stats_part (optional): [B][hdrsky_da_conv_stats_nparts(H,W)][2][Cout] InstanceNorm partial sums of y in the layout the
 * plain conv emits, for hdrsky_norm_apply - the commented-out distortion-aware res blocks of generator.py:14,18. */
int hdrsky_da_conv_stats_nparts(int H, int W); /* [host] */
/* row_lo / spans (optional, BF16 mode): the offsets depend on the image row only, so a group of consecutive 64-pixel tiles
 * (row-major) samples a few consecutive source rows.  For group sizes G = 1, 2, 4, 8, 16 (level l = log2 G): row_lo =
 * device int32 [5][ceil(H*W/64)], first source row of group g at [l][g]; spans = HOST int32 [5], rows from there that cover
 * every group's samples (both from hdrsky_da_sample_table on the host: kernels.da_row_lo).  With them a workgroup stages the
 * rows of its group once in LDS (bf16) and gathers every corner from there instead of from L2; NULL, BF16X3, or spans that
 * do not fit LDS: the corners come from global memory (fp32 sources). */
int hdrsky_da_conv2d_fwd(const float* x, const void* w_hi, const void* w_lo, const float* bias, const float* offs,
                         const int* row_lo, const int* spans, int B, int H, int W, int Cin, int Cout, int ksize, int compute,
                         float* y, float* stats_part, void* stream);
/* Backward building blocks of the distortion-aware conv (tf.GradientTape through distortion_aware_ops.py:62-121):
 * with G = hdrsky_da_gather(x) [B,H,W,k*k*C] the layer is a 1x1 conv of G, so dW = hdrsky_conv2d_wgrad(1x1; G, dY),
 * dG = hdrsky_conv2d_fwd(1x1 with the transposed kernel; dY) and dx = hdrsky_da_scatter(dG) (dx zeroed; fp32 atomics). */
int hdrsky_da_gather(const float* x, const float* offs, int B, int H, int W, int C, int ksize, float* G, void* stream);
int hdrsky_da_scatter(const float* dG, const float* offs, int B, int H, int W, int C, int ksize, float* dx, void* stream);
/* The gathered operand as a bf16 tensor G [B,H,W,k*k*C] - what the fused kernels feed the matrix cores, written once
 * (single-product mode).  With it the layer is distortion_aware_ops.py:107-121 literally: y = hdrsky_conv2d_fwd(1x1 over
 * k*k*C channels; G) with the k x k filter's own packed image (same k-step order), dW = hdrsky_conv2d_wgrad(1x1; G, dY), and
 * the data gradient is the same pair on the transposed table.  Positions: offs (device, hdrsky_da_offsets) = the forward's
 * four corners; or offs = NULL and gidx / gw [H*W][k*k][km] = a sample table (hdrsky_da_conv2d_dgrad's, km = 8: then x is dY
 * and G the operand of the 1x1 conv with the transpose_flip image).  x: fp32, or bf16 with x_bf16 != 0.  C % 8 == 0. */
int hdrsky_da_gather_bf16(const void* x, int x_bf16, const float* offs, const int* gidx, const float* gw, int km, int B, int H, int W, int C, int ksize, void* G, void* stream);
/* The matmul on that operand (distortion_aware_ops.py:117-121: tf.matmul(gathered, kernel) + bias), a 1x1 convolution over
 * K = k*k*C channels of a FINAL bf16 NHWC tensor: y [B*HW][N] = A [B*HW][K] (bf16) x W + bias, W = the hi plane of the
 * hdrsky_conv_pack_weights image of the k x k filter (or of its transpose_flip image: data gradient on the transposed table).
 * Both operands travel global -> LDS by LDS-DMA through a four-stage ring (csrc/gemm_1x1.hip); hdrsky_conv2d_fwd takes the
 * same call as a 1x1 conv, in power-of-two channel groups, 2-3x slower.  stats_part (optional): InstanceNorm partials
 * [B][hdrsky_gemm1x1_stats_nparts(HW)][2][N] of the fp32 results, for hdrsky_norm_apply / hdrsky_in_affine.
 * hdrsky_gemm1x1_supported: HW % 128 == 0 (whole 128-pixel tiles per sample), K % 64 == 0, N % 32 == 0. */
int hdrsky_gemm1x1_supported(int HW, int K, int N); /* [host] */
int hdrsky_gemm1x1_stats_nparts(int HW); /* [host] */
int hdrsky_gemm1x1_bf16(const void* A, const void* w_hi, const float* bias, int B, int HW, int K, int N, void* y, int y_bf16, float* stats_part, void* stream);
/* [host] The forward's sample table of an H x W map: per (pixel oy*W+ox, tap) the four bilinear corners as source pixel
 * indices (row-major, -1 = zero padding) and weights, [H*W][k*k][4] each - exactly what hdrsky_da_conv2d_fwd gathers
 * (distortion_aware_ops.py:62-106 in float32).  offs: HOST copy of hdrsky_da_offsets. */
int hdrsky_da_sample_table(const float* offs, int H, int W, int ksize, int* idx, float* w);
/* Data gradient of the distortion-aware conv (tape through distortion_aware_ops.py:62-121) WITHOUT the k*k-fold tensor and
 * without atomics: dx[q][c] = sum_t sum_f (sum_{(p,w) in L(q,t)} w dy[p][f]) W[t][c][f] - the forward kernel run on the
 * transposed sample table.  gidx / gw: device [H*W][k*k][8] (source pixel, -1 = none / weight), tap order of wT_*;
 * wT_* = hdrsky_conv_pack_weights(kernel viewed [k,k,C,F], ..., transpose_flip=1) (Cin = F filters, Cout = C).
 * row_lo / spans: as for hdrsky_da_conv2d_fwd, for the rows of dy the table's sources of a tile lie in. */
int hdrsky_da_conv2d_dgrad(const float* dy, const void* wT_hi, const void* wT_lo, const int* gidx, const float* gw,
                           const int* row_lo, const int* spans, int B, int H, int W, int F, int C, int ksize, int compute, float* dx,
                           void* stream);
/* Kernel gradient of the distortion-aware conv (tape through distortion_aware_ops.py:62-121) with the gather taken from LDS
 * (BF16 mode): dw [k*k*C, F] += G(x)^T dy, db [F] += column sums of dy (db may be NULL); the [B,H,W,k*k,C] operand never
 * exists.  dy fp32 or (dy_bf16) bf16; offs = device offsets; row_lo / spans as for hdrsky_da_conv2d_fwd.  A workgroup
 * stages the source rows of a group of tiles of one sample once and walks (taps, tiles); partial slabs per (sample, group)
 * in ws (hdrsky_da_conv2d_wgrad_ws_bytes bytes; 0 = layer not supported here) are added to dw by a second launch in a
 * fixed order: deterministic.  HDRSKY_EUNSUPPORTED: use hdrsky_conv2d_wgrad_multi with the job's da_* fields instead. */
size_t hdrsky_da_conv2d_wgrad_ws_bytes(const int* row_lo, const int* spans, int B, int H, int W, int C, int F, int ksize); /* [host] */
int hdrsky_da_conv2d_wgrad(const float* x, const void* dy, int dy_bf16, const float* offs, const int* row_lo, const int* spans,
                           int B, int H, int W, int C, int F, int ksize, float* dw, float* db, void* ws, size_t ws_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * Sample-resident 3x3 convolution with the InstanceNormalization around it fused in (csrc/res_conv.hip).
 * Replaces, for the 8 x 32-pixel maps of the 32 x 128 configuration, one half of generator.resBlock.call
 * (generator.py:26-35: `ops.conv2d` -> tfa InstanceNormalization -> tf.nn.leaky_relu(0.1) | -> tf.add(identity)) in
 * ONE launch, and in the backward pass (train.py:402, tape.gradient through those lines) the data gradient of the conv
 * together with the gradient through the normalisation (and activation) in front of it.  Tensors that travel between
 * these launches are FINAL bf16 activations [B,8,32,C] (the values the matrix cores consume; compute mode
 * HDRSKY_BF16 only) - the residual stream and its gradient stay fp32.
 * ---------------------------------------------------------------------------------------- */
#define HDRSKY_RC_FWD 0
#define HDRSKY_RC_BWD 1
typedef struct hdrsky_resconv_args {
  int32_t B, Cin, Cout;  /* x [B,8,32,Cin] bf16 (Cin 64 or 128), outputs [B,8,32,Cout], Cout % 16 == 0 */
  int32_t mode;          /* HDRSKY_RC_FWD | HDRSKY_RC_BWD */
  float slope;           /* leaky slope of the activation behind the norm (1 = none, 0 = relu) */
  float eps;             /* variance epsilon (1e-3 for tfa InstanceNormalization) */
  const void* x;         /* conv operand, bf16, final values (no transform).  NULL (BWD only): no convolution, the
                          * "conv result" is zero and `res` alone is differentiated through the norm */
  const void* w;         /* hi plane of hdrsky_conv_pack_weights(3,3,Cin,Cout) (transpose_flip=1 image for BWD) */
  const float* bias;     /* FWD: the conv bias [Cout] or NULL - accepted and NOT read: a per-channel constant in front of
                          * an InstanceNormalization cancels exactly (its gradient is zero for the same reason) */
  const float* gamma;    /* [Cout] scale / offset of the norm: FWD required; BWD when xhat_in != NULL */
  const float* beta;
  const float* res;      /* fp32 [B,8,32,Cout] or NULL.  FWD: added after the activation (identity branch);
                          * BWD: added to the conv result before the norm backward (gradient of the identity branch) */
  const void* xhat_in;   /* BWD: bf16 normalised pre-activation (conv - mean) * rstd saved by the forward launch of the
                          * norm being differentiated, or NULL = no norm behind this data gradient */
  const float* inv_in;   /* BWD: rstd [B,Cout] saved by that forward launch */
  void* y_bf16;          /* FWD: act(norm(conv + bias)) + res as bf16.  BWD: gradient w.r.t. the conv output in front of
                          * the differentiated norm (or the plain data gradient when xhat_in == NULL).  Nullable */
  float* y_f32;          /* FWD: the same value as fp32 (the residual stream).  BWD: conv result + res (the stream's
                          * gradient, before the norm backward).  Nullable */
  void* xhat_out;        /* FWD: bf16 normalised pre-activation for the backward pass.  Nullable */
  float* inv_out;        /* FWD: rstd [B,Cout].  Nullable */
  float* dgb;            /* BWD with a norm: per-sample [B][2][Cout] (d gamma, d beta) terms; reduce over B with
                          * hdrsky_dgb_reduce.  Nullable */
} hdrsky_resconv_args;
/* 1 when hdrsky_resconv handles this layer geometry (8 x 32 pixels, 3x3, Cin 64|128, Cout % 16 == 0). [host] */
int hdrsky_resconv_supported(int H, int W, int Cin, int Cout, int KH, int KW);
int hdrsky_resconv(const hdrsky_resconv_args* args, void* stream);
/* Per-channel sums over the batch of `nlayers` per-sample tables in one launch, batch order fixed (bit-reproducible): the
 * (d gamma, d beta) of norm layers from hdrsky_resconv's `dgb` ([B][2][C]: d gamma, d beta) or from hdrsky_norm_act_bwd's
 * `sums` ([B][2][C]: d beta, d gamma).  table: device array [nlayers][4] of int64
 * {part ptr ([B][2][C]), dst0 ptr ([C], += sum_b part[b][0], nullable), dst1 ptr ([C], += sum_b part[b][1], nullable), C}. */
int hdrsky_dgb_reduce(const void* table, int nlayers, int B, void* stream);
/* y (bf16) = round-to-nearest-even(x) for n contiguous floats, n % 8 == 0: entry of a bf16 activation chain. */
int hdrsky_to_bf16(const float* x, void* y, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HDRSKY_H */
