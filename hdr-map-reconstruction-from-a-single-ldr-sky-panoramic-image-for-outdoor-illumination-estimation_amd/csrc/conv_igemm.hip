// Implicit-GEMM NHWC convolution on gfx950 matrix cores (v_mfma_f32_16x16x32_bf16).
//
// One workgroup (4 waves) owns an output tile of TH x TW pixels of one sample and BN output
// channels.  The input halo patch of that tile is staged ONCE into LDS as bf16 channel-chunk
// planes ([Cin/8][halo pixel] of 16-byte units) with the producer's normalisation + activation
// (and, for the resize-deconv, the 2x bilinear resize; for stride-2 dgrad, the zero stuffing)
// applied on the way in.  The im2col matrix is never materialised: an MFMA A fragment for tap
// (ky,kx) is a ds_read_b128 of the halo plane at a shifted pixel address.  Weights are
// pre-packed into the B-fragment image [kstep][4][Npad][8] bf16 and streamed through a
// double-buffered LDS ring, KC k-steps per barrier.
//
// Replaces tf.nn.conv2d + bias_add (ops.py:41-42), resize+conv (ops.py:121-124), Keras Conv2D
// (discriminator.py:11-13, sunrad_net.py:12-14), vgg16.conv2d (vgg16.py:32-36) and, through
// transposed/flipped packed filters, their data gradients.
#include <cstdio>
#include <cstdlib>

#include <type_traits>

#include <atomic>

#include "common.h"
#include "hooks.h"

namespace {

constexpr int kc_for(int bn) { return bn <= 32 ? 8 : 4; }  // k-steps (of 32) per B chunk / barrier

struct ConvKArgs {
  const float* x;
  const uint4* whi;
  const uint4* wlo;
  const float* bias;
  const float* in_scale;
  const float* in_shift;
  const float* in_part;
  const float* in_gamma;
  const float* in_beta;
  const float* residual;
  float* y;
  float* stats;
  int B, H, W, Cin, Ho, Wo, Cout, Npad;
  int KH, KW, stride, pad_t, pad_l, upsample, dilate, Hc, Wc;
  int in_mode, ss_bstride, in_nparts;
  float in_eps, in_inv_count, in_slope, out_slope;
  int final_relu, want_stats;
  int tiles_x, tiles_y, nblocks;
  int cgs, log2nq, ngroups, ksg, log2cbg, ntaps, kzero;
  int HT, WT, NPIX, NPIXP, wt_magic, kw_magic;
  int off_alo, off_b, off_ss, off_tap, off_stat, off_ktab, off_out;
  unsigned long long* stamps;  // debug: per-workgroup phase time stamps (null in production)
  int x_bf16, y_bf16;          // activation storage (hdrsky_conv_desc)
  int res_mode; float mask_slope;
  // xb_out (optional, single-product mode, plain staging): the TRANSFORMED operand act(norm(x)) as a bf16 NHWC tensor of x's shape -
  // exactly the values staged for the matrix cores, written by the workgroups of output-channel block 0, every input pixel by
  // the tile that owns it.  It is the operand of the layer's weight gradient (conv_wgrad2_kernel), which needed a launch of its
  // own (hdrsky_act_bf16) to produce it.
  unsigned short* xb_out;
  // stride-2 data gradient by output phases (phase != 0): workgroup = a tile of ONE of the four (oy & 1, ox & 1) phases of
  // the output; KH / KW are then the per-phase tap counts ceil(K / 2), KHf / KWf the full (flipped) filter, pad_t / pad_l
  // the full conv's K - 1 - pad, H / W (= Hc / Wc) the un-stuffed gradient
  int phase, KHf, KWf;
  // PAIRED launch (hdrsky_conv2d_fwd_pair): two layers of identical geometry in one grid - samples [0, gsplit) use the first
  // parameter set (whi, wlo, bias, in_gamma, in_beta, residual), samples [gsplit, B) the second (the *2 pointers).  x_shared:
  // both halves read the same gsplit input samples (two layers on one input).  gsplit == 0: an ordinary launch.
  const uint4* whi2; const uint4* wlo2; const float* bias2; const float* in_gamma2; const float* in_beta2; const float* residual2;
  int gsplit, x_shared;
};

// Packed-filter k-step of phase tap `tapp` (ky', kx' of the per-phase KH x KW grid), channel block cb: the full filter's
// tap (ay + 2 ky', ax + 2 kx'); taps beyond an odd filter's edge read the packed image's all-zero k-step.
__device__ __forceinline__ int phase_kp(const ConvKArgs& a, int tapp, int cbk, int cin32, int ph) {
  const int ay = (a.pad_t - (ph >> 1)) & 1, ax = (a.pad_l - ph) & 1;
  const int kyp = (tapp * a.kw_magic) >> 16;
  const int ky = ay + 2 * kyp, kx = ax + 2 * (tapp - kyp * a.KW);
  return (ky < a.KHf && kx < a.KWf) ? (ky * a.KWf + kx) * cin32 + cbk : a.kzero;
}

// load 8 consecutive channels and apply the producer's affine + leaky activation
__device__ __forceinline__ void load8_xf(const float* __restrict__ p, const float* sc, const float* sh, bool xf,
                                         float slope, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  if (xf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = leaky(v[j] * sc[j] + sh[j], slope);
  } else if (slope != 1.f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = leaky(v[j], slope);
  }
}

#ifndef HDRSKY_STAMP_TID
#define HDRSKY_STAMP_TID 0      // the thread that writes the phase stamps (diagnostic builds: another wave's first lane)
#endif
#define HDRSKY_STAMP(k)                                                               \
  if (a.stamps != nullptr && threadIdx.x == HDRSKY_STAMP_TID) {                       \
    a.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();            \
    if ((k) == 0 || (k) == 5) a.stamps[(size_t)blockIdx.x * 8 + 6 + ((k) ? 1 : 0)] = __builtin_amdgcn_s_memrealtime(); \
  }

// KC k-steps of MFMA on one LDS-resident B chunk; TAIL=true guards each k-step against the end of the
// reduction (only the last chunk can be partial)
template <int MI, int NI, int BN, int KC, int BPLANES, int BITEMS, bool NARROW, bool PRECISE, bool TAIL>
__device__ __forceinline__ void compute_chunk(const ConvKArgs& a, const unsigned char* smem, const uint4* sB,
                                              const int* sTap, int ch, int kq, const int (&abase)[MI],
                                              const int (&bbase)[NI], f32x4_t (&acc)[MI][NI]) {
  const int buf = ch & 1;
  const unsigned char* bh = reinterpret_cast<const unsigned char*>(sB + (buf * BPLANES) * BITEMS);
  const unsigned char* bl = reinterpret_cast<const unsigned char*>(sB + (buf * BPLANES + (PRECISE ? 1 : 0)) * BITEMS);
#pragma unroll
  for (int ksl = 0; ksl < KC; ++ksl) {
    const int ks = ch * KC + ksl;
    if (TAIL && ks >= a.ksg) break;
    // operand-plane offset of this k-step: (ky*WT + kx) pixels [+ channel block]; division by KW is a
    // multiply-shift (exact for tap < 64), scalar in the WIDE case
    int aoff;
    if (NARROW) {
      const int tap = min(ks * 4 + kq, a.ntaps - 1);
      const int ky = (tap * a.kw_magic) >> 16;
      aoff = (ky * a.WT + (tap - ky * a.KW)) * 16;
    } else {
      const int tap = min(ks >> a.log2cbg, a.ntaps - 1), cb = ks & ((1 << a.log2cbg) - 1);
      const int ky = (tap * a.kw_magic) >> 16;
      aoff = (ky * a.WT + (tap - ky * a.KW) + cb * 4 * a.NPIXP) * 16;
    }
    uint4 ah[MI], al[MI], wh[NI], wl[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ah[mi] = *reinterpret_cast<const uint4*>(smem + abase[mi] + aoff);
      if (PRECISE) al[mi] = *reinterpret_cast<const uint4*>(smem + a.off_alo + abase[mi] + aoff);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      wh[ni] = *reinterpret_cast<const uint4*>(bh + bbase[ni] + ksl * (4 * BN * 16));
      if (PRECISE) wl[ni] = *reinterpret_cast<const uint4*>(bl + bbase[ni] + ksl * (4 * BN * 16));
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (PRECISE) {
          acc[mi][ni] = mfma16(al[mi], wh[ni], acc[mi][ni]);
          acc[mi][ni] = mfma16(ah[mi], wl[ni], acc[mi][ni]);
        }
        acc[mi][ni] = mfma16(ah[mi], wh[ni], acc[mi][ni]);
      }
  }
}

// B chunk `chunk_` (k-step index is wave-uniform; clamped at the end of the reduction: surplus k-steps are fetched but
// never multiplied) by LDS-DMA straight into ring buffer buf_ - no registers in between (a register copy in flight across
// the long operand staging / the MFMA phase was parked in scratch by the compiler); completion: s_waitcnt vmcnt(0) before
// the barrier that publishes it
#define HDRSKY_DMA_B(chunk_, buf_)                                                                        \
  _Pragma("unroll") for (int j = 0; j < BPT; ++j) {                                                      \
    const int ks_ = min((chunk_) * KC + j * RPP + bsub, a.ksg - 1);                                       \
    int kp_;                                                                                              \
    if (NARROW) kp_ = ks_;                                                                                \
    else if (PH) kp_ = phase_kp(a, ks_ >> a.log2cbg, (g << a.log2cbg) + (ks_ & ((1 << a.log2cbg) - 1)), cin32, ph); \
    else kp_ = (ks_ >> a.log2cbg) * cin32 + (g << a.log2cbg) + (ks_ & ((1 << a.log2cbg) - 1));           \
    const size_t src_ = (size_t)(kp_ * 4) * a.Npad + boff;                                                \
    glds16(whi + src_, sb_lds + (((buf_) * BPLANES + 0) * BITEMS + wave_base + j * NT) * 16);           \
    if (PRECISE) glds16(wlo + src_, sb_lds + (((buf_) * BPLANES + 1) * BITEMS + wave_base + j * NT) * 16); \
  }
// 8-wave variants are compiled for 4 waves per SIMD (<= 128 VGPRs; they need 77-106 and no scratch): two workgroups
// then fit a CU, which is what lets kernels of the other streams of the training step overlap with this one.
// PH: the stride-2 data gradient by output phases (ConvKArgs::phase; its own instantiations: the extra scalars of the
// phase arithmetic pushed the 8-wave direct-B variants, which sit at their 128-VGPR budget, into scratch)
// EMIT: the instantiations that also write the transformed operand (ConvKArgs::xb_out) - their own, because the extra address
// registers of the staging pushed the 8-wave variants at their 128-VGPR budget into a spill around the k loop (+3.5 % on the
// training step when every launch paid for it; as instantiations only the nine launches per step that emit do)
// PAIR: the instantiations of the paired launches (ConvKArgs::gsplit) - their own for the reason EMIT's are: the second parameter
// set's scalars pushed the 8-wave direct-B variants at their 128-VGPR budget into a spill around the k loop when every launch
// carried them (four plain instantiations went from 0 to 36 B of scratch); only the tiles the paired layers take are instantiated
template <int WM, int WN, int MI, int NI, int TW, bool NARROW, bool PRECISE, bool DB, bool PH, bool EMIT = false, bool PAIR = false, bool UP = false>
#ifndef HDRSKY_EARLY_B
#define HDRSKY_EARLY_B 1       // 0: the filter window is filled behind the staging barrier (A/B builds)
#endif
#ifndef HDRSKY_EPI_DIRECT
#define HDRSKY_EPI_DIRECT 1    // 0: every variant through the LDS tile (A/B builds)
#endif
#ifndef HDRSKY_DB_MINW
#define HDRSKY_DB_MINW 4      // waves per SIMD the 8-wave direct-B instantiations are compiled for (register budget 512 / that)
#endif
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN == 8 && !PRECISE) ? (DB ? HDRSKY_DB_MINW : 2) : 1) conv_igemm_kernel(const ConvKArgs a) {
  constexpr int NW = WM * WN;                // waves per workgroup (4 or 8)
  constexpr int NT = NW * 64;
  constexpr int BM = WM * MI * 16;
  constexpr int BN = WN * NI * 16;
  constexpr int TH = BM / TW;
  constexpr int FPR = TW / 16;               // M fragments per tile row
  constexpr int KC = kc_for(BN);
  constexpr int BITEMS = KC * 4 * BN;        // uint4 per B chunk plane
  constexpr int BPT = BITEMS / NT;           // per thread
  constexpr int BPLANES = PRECISE ? 2 : 1;
  constexpr int BNP = BN + 4;                // padded row of the epilogue tile
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  // Epilogue straight from the accumulators (single-product direct-B variants): the MFMAs run with the operands swapped -
  // D[cout][pixel] instead of D[pixel][cout] - so that a lane holds FOUR CONSECUTIVE CHANNELS of one pixel: bias, statistics,
  // activation, residual / mask and the 16-byte (fp32) or 8-byte (bf16) store happen on the registers, with no LDS tile, no
  // barrier in front of it and no wait for the slowest wave of the workgroup.  Every output value is the same sum in the same
  // order (bit-identical y); the statistics partials add their pixels in another order than the LDS-tile epilogue.  The
  // fp32-class mode keeps the LDS tile (its statistics, and with them the long fits of the parity tests, stay bit-stable).
  constexpr bool DIRECT = DB && !PRECISE && (HDRSKY_EPI_DIRECT != 0);
  static_assert(BITEMS % NT == 0 && BPT >= 1, "B chunk must tile the block");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* sAhi = reinterpret_cast<uint4*>(smem);
  uint4* sAlo = reinterpret_cast<uint4*>(smem + a.off_alo);
  uint4* sB = reinterpret_cast<uint4*>(smem + a.off_b);  // [2 bufs][hi,lo][BITEMS]
  float* sScale = reinterpret_cast<float*>(smem + a.off_ss);
  float* sShift = sScale + a.Cin;
  int* sTap = reinterpret_cast<int*>(smem + a.off_tap);
  float* sStat = reinterpret_cast<float*>(smem + a.off_stat);
  int2* sKtab = reinterpret_cast<int2*>(smem + a.off_ktab);   // direct-B loop: per k-step {A offset of the next step, filter refill offset}

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kq = lane >> 4, lr = lane & 15;

  // XCD-aware order: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its
  // own L2); remap so that each XCD owns a CONTIGUOUS range of logical tiles (= a few whole samples) and
  // the activations are pulled through the fabric once, not once per XCD.  Bijective for any grid size.
  int bid;
  {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int qd = nwg >> 3, rm = nwg & 7;
    bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
  }
  const int nb = bid % a.nblocks; bid /= a.nblocks;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  // Output phases of a stride-2 data gradient (a.phase): dx[2m + p] = sum_t dy[m + off_p + t] * w'[a_p + 2t] with
  // a_p = (pad' - p) & 1, off_p = (p + a_p - pad') / 2 (pad' = K - 1 - pad of the forward conv, w' the flipped filter) -
  // a stride-1 conv of the UN-stuffed gradient with every other tap, i.e. a quarter of the zero-stuffed form's products
  // (the same non-zero products in the same order: bit-identical).  The four phases of a sample are neighbours in the
  // workgroup order (they read the same gradient tile).
  int b = bid, ph = 0, pad_t = a.pad_t, pad_l = a.pad_l;     // ph = 2 * (oy & 1) + (ox & 1) of this workgroup's phase
  if (PH) {
    b = bid >> 2; ph = bid & 3;
    const int py = ph >> 1, px = ph & 1;
    pad_t = -((py + ((a.pad_t - py) & 1) - a.pad_t) >> 1); pad_l = -((px + ((a.pad_l - px) & 1) - a.pad_l) >> 1);   // (even numerators: exact)
  }
  // paired launch: the second half of the batch runs on the second layer's parameters (workgroup-uniform scalar selects)
  const bool g1 = PAIR && b >= a.gsplit;
  const uint4* const whi = g1 ? a.whi2 : a.whi;
  const uint4* const wlo = g1 ? a.wlo2 : a.wlo;
  const float* const bias = g1 ? a.bias2 : a.bias;
  const float* const in_gamma = g1 ? a.in_gamma2 : a.in_gamma;
  const float* const in_beta = g1 ? a.in_beta2 : a.in_beta;
  const float* const residual = g1 ? a.residual2 : a.residual;
  const int bx = (g1 && a.x_shared) ? b - a.gsplit : b;      // sample of the INPUT tensor (and of its transform tables)
  const int br = g1 ? b - a.gsplit : b;                      // sample inside the group's own residual tensor
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = nb * BN;    // (phase mode: coordinates on the phase's own grid)
  const int iy0 = oy0 * a.stride - pad_t, ix0 = ox0 * a.stride - pad_l;

  HDRSKY_STAMP(0)
  // ---- prologue: tap offset table + input transform tables (identity when in_mode == NONE) ----
  if (!DB) { for (int t = tid; t < a.ntaps; t += NT) sTap[t] = (t / a.KW) * a.WT + (t % a.KW); }   // (the direct-B loop reads sKtab instead)
  if (DB && !NARROW) {
    // k-step table of the direct-B main loop (channel group 0; a group only shifts the filter base): entry ks holds the
    // LDS byte offset of the A fragments of step ks+1 and the packed-filter element offset of step ks+DPF, both clamped
    // to the last step.  The loop then needs no scalar index arithmetic: one LDS read per DPF steps + v_readlane.
    constexpr int DPFT = (NI == 1) ? 8 : 4;
    const int cbm = (1 << a.log2cbg) - 1, cin32t = a.Cin >> 5;
    const int nent = (a.ksg + DPFT - 1) / DPFT * DPFT;
    for (int i = tid; i < nent; i += NT) {
      const int ka = min(i + 1, a.ksg - 1), kb = min(i + DPFT, a.ksg - 1);
      const int tap = ka >> a.log2cbg, cb = ka & cbm;
      const int ky = (tap * a.kw_magic) >> 16;
      const int aoff = (ky * a.WT + (tap - ky * a.KW) + cb * 4 * a.NPIXP) * 16;
      const int kp = PH ? phase_kp(a, kb >> a.log2cbg, kb & cbm, cin32t, ph) : (kb >> a.log2cbg) * cin32t + (kb & cbm);
      sKtab[i] = int2{aoff, kp * 4 * a.Npad};
    }
  }
  if (a.in_mode == HDRSKY_IN_AFFINE) {
    for (int c = tid; c < a.Cin; c += NT) {
      sScale[c] = a.in_scale[bx * a.ss_bstride + c];
      sShift[c] = a.in_shift[bx * a.ss_bstride + c];
    }
  } else if (a.in_mode == HDRSKY_IN_PARTIALS) {
    for (int c = tid; c < a.Cin; c += NT) {
      float s, ss;
      in_partial_sums(a.in_part + (size_t)bx * a.in_nparts * 2 * a.Cin + c, a.in_nparts, a.Cin, s, ss);
      const float mean = s * a.in_inv_count;
      const float var = fmaxf(ss * a.in_inv_count - mean * mean, 0.f);
      const float inv = in_gamma[c] / sqrtf(var + a.in_eps);
      sScale[c] = inv;
      sShift[c] = in_beta[c] - mean * inv;
    }
  } else if (a.in_slope != 1.f || NARROW || UP) {
    for (int c = tid; c < a.Cin; c += NT) { sScale[c] = 1.f; sShift[c] = 0.f; }
  }
  // an identity input on the plain staging path (a final activation: the VGG16 chain, the data gradients) reads neither table
  // before the barrier that ends the staging: no barrier here
  const bool tables_read = !(a.in_mode == HDRSKY_IN_NONE && a.in_slope == 1.f && !NARROW && !UP && DB);
  if (tables_read) __syncthreads();
  HDRSKY_STAMP(1)

  // ---- per-lane fragment bases ---------------------------------------------------------------
  int abase[MI];  // byte offset of this lane's A row (pixel) inside a plane (+ its k-quarter plane)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int f = wm * MI + mi;
    const int fr = f / FPR, fc = f % FPR;
    const int pix = (fr * a.stride) * a.WT + (fc * 16 + lr) * a.stride;
    abase[mi] = (NARROW ? pix : (kq * a.NPIXP + pix)) * 16;
  }
  int bbase[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bbase[ni] = (kq * BN + (wn * NI + ni) * 16 + lr) * 16;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nchunks = (a.ksg + KC - 1) / KC;
  constexpr int RPP = NT / (4 * BN);  // k-steps covered by one pass of the block over a B chunk
  const int bsub = __builtin_amdgcn_readfirstlane(tid / (4 * BN));
  const unsigned sb_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(smem + a.off_b);
  const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
  const int boff = ((tid % (4 * BN)) / BN) * a.Npad + n0 + (tid % BN);  // (q, n) part of the packed index
  const int cin32 = a.Cin >> 5;
  const float slope = a.in_slope;
  const bool xf_identity = a.in_mode == HDRSKY_IN_NONE && slope == 1.f;   // staging then skips the affine + activation

  // direct-B loop: the register window of filter fragments (DPF k-steps x NI).  In the single-product mode its first fill is
  // issued BEFORE the group's operand staging: the L2 round trip of the filter rows runs under the staging instead of behind
  // the barrier that ends it (32 registers live across the staging; the fp32-class mode's 64 would not fit beside it).
  constexpr int DPF = (NI == 1) ? 8 : 4;
  constexpr bool EARLY_B = DB && !PRECISE && !UP && !EMIT && (MI * NI < 8 || NW == 4) && (HDRSKY_EARLY_B != 0);     // (EMIT's owner bookkeeping and the 32 accumulator registers of the 8-wave 128 px x 128 ch tile spill beside the window)
  uint4 bqh[DB ? DPF : 1][NI], bql[DB ? DPF : 1][NI];
  const unsigned kstride = 4u * (unsigned)a.Npad;   // uint4 elements per k-step; 32-bit offsets: the packed filter is < 2^32 elements
  const uint4* const wlh = whi + (size_t)kq * a.Npad + n0 + (wn * NI) * 16 + lr;
  const uint4* const wll = PRECISE ? wlo + (size_t)kq * a.Npad + n0 + (wn * NI) * 16 + lr : nullptr;
  auto kp_of = [&](int g, int ks) __attribute__((always_inline)) {
    const int kc = min(ks, a.ksg - 1);
    if (!NARROW && PH) return phase_kp(a, kc >> a.log2cbg, (g << a.log2cbg) + (kc & ((1 << a.log2cbg) - 1)), cin32, ph);
    return NARROW ? kc : (kc >> a.log2cbg) * cin32 + (g << a.log2cbg) + (kc & ((1 << a.log2cbg) - 1));
  };
  auto fill_b = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < DPF; ++j) {
      const unsigned o = (unsigned)kp_of(g, j) * kstride;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bqh[j][ni] = wlh[o + ni * 16];
        if (PRECISE) bql[j][ni] = wll[o + ni * 16];
      }
    }
  };

  for (int g = 0; g < a.ngroups; ++g) {
    if (g > 0 && !DB) __syncthreads();  // everyone finished reading the previous group's planes
    if constexpr (EARLY_B) fill_b(g);

    // LDS-ring variant: first B chunk goes in flight (LDS-DMA into ring buffer 0) before the (long) A staging
    if (!DB) { HDRSKY_DMA_B(0, 0) }

    // ---- stage the halo patch of this channel group into LDS (branch-free per item) ----------
    if (NARROW) {
      for (int p = tid; p < a.NPIX; p += NT) {
        const int hy = (int)(((unsigned)p * (unsigned)a.wt_magic) >> 24);
        const int hx = p - hy * a.WT;
        const int cy = iy0 + hy, cx = ix0 + hx;
        const bool ok = cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
        const float* src = a.x + ((size_t)(bx * a.H + (ok ? cy : 0)) * a.W + (ok ? cx : 0)) * a.Cin;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = src[j < a.Cin ? j : 0];
          const float u = leaky(t * sScale[j < a.Cin ? j : 0] + sShift[j < a.Cin ? j : 0], slope);
          v[j] = (ok && j < a.Cin) ? u : 0.f;
        }
        uint4 hi, lo;
        pack8<PRECISE>(v, hi, lo);
        sAhi[p] = hi;
        if (PRECISE) sAlo[p] = lo;
      }
    } else {
      const int nq = 1 << a.log2nq;
      const int nitems = a.NPIX << a.log2nq;
      const float* xb = a.x + (size_t)bx * a.H * a.W * a.Cin + g * a.cgs;
      // NT % nq == 0, so a thread always stages the same 8-channel chunk: its affine lives in registers
      const int qt = tid & (nq - 1);
      float sc8[8], sh8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc8[j] = sScale[g * a.cgs + qt * 8 + j]; sh8[j] = sShift[g * a.cgs + qt * 8 + j]; }
      if constexpr (UP) {     // the resize-fused staging: instantiations of their own (its 32 source registers per item beside the
                              // filter window and the plain path's batch would not fit the 128-register budget: dispatch_tile_up)
        for (int i = tid; i < nitems; i += NT) {
          const int p = i >> a.log2nq, q = i & (nq - 1);
          const int hy = (int)(((unsigned)p * (unsigned)a.wt_magic) >> 24);
          const int hx = p - hy * a.WT;
          const int cy = iy0 + hy, cx = ix0 + hx;
          const bool ok = cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
          // tf.image.resize BILINEAR, half-pixel centres: src = (dst+0.5)*0.5-0.5
          const float sy = (cy + 0.5f) * 0.5f - 0.5f, sx = (cx + 0.5f) * 0.5f - 0.5f;
          const float fy = floorf(sy), fx = floorf(sx);
          const int ylo = min(max((int)fy, 0), a.H - 1), yhi = max(min((int)ceilf(sy), a.H - 1), 0);
          const int xlo = min(max((int)fx, 0), a.W - 1), xhi = max(min((int)ceilf(sx), a.W - 1), 0);
          const float ly = sy - fy, lx = sx - fx;
          const float* s00 = xb + ((size_t)ylo * a.W + xlo) * a.Cin + q * 8;
          const float* s01 = xb + ((size_t)ylo * a.W + xhi) * a.Cin + q * 8;
          const float* s10 = xb + ((size_t)yhi * a.W + xlo) * a.Cin + q * 8;
          const float* s11 = xb + ((size_t)yhi * a.W + xhi) * a.Cin + q * 8;
          float4 t[8];
          t[0] = *reinterpret_cast<const float4*>(s00); t[1] = *reinterpret_cast<const float4*>(s00 + 4);
          t[2] = *reinterpret_cast<const float4*>(s01); t[3] = *reinterpret_cast<const float4*>(s01 + 4);
          t[4] = *reinterpret_cast<const float4*>(s10); t[5] = *reinterpret_cast<const float4*>(s10 + 4);
          t[6] = *reinterpret_cast<const float4*>(s11); t[7] = *reinterpret_cast<const float4*>(s11 + 4);
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float sc = sc8[j], sh = sh8[j];
            const int w = j >> 2, e = j & 3;
            const float tl = leaky(((const float*)&t[0 + w])[e] * sc + sh, slope);
            const float tr = leaky(((const float*)&t[2 + w])[e] * sc + sh, slope);
            const float bl = leaky(((const float*)&t[4 + w])[e] * sc + sh, slope);
            const float br = leaky(((const float*)&t[6 + w])[e] * sc + sh, slope);
            const float top = tl + (tr - tl) * lx;
            const float bot = bl + (br - bl) * lx;
            v[j] = ok ? top + (bot - top) * ly : 0.f;
          }
          uint4 hi, lo;
          pack8<PRECISE>(v, hi, lo);
          sAhi[q * a.NPIXP + p] = hi;
          if (PRECISE) sAlo[q * a.NPIXP + p] = lo;
        }
      } else {
        // plain / zero-stuffed operand: UNR items per thread, all loads issued before any use
        constexpr int UNR = 4;
        const int dummy = a.NPIXP - 1;  // pad slot of plane 0: never read
        const int dsh = a.dilate == 2 ? 1 : 0;
        // xb_out: this tile owns the input pixels under its own output block (TH x TW outputs x stride: a partition of the image)
        const bool emit_xb = EMIT && !PRECISE && nb == 0;
        const int own_y0 = oy0 * a.stride, own_y1 = (oy0 + TH) * a.stride, own_x0 = ox0 * a.stride, own_x1 = (ox0 + TW) * a.stride;
        unsigned short* xbo = a.xb_out + (size_t)b * a.H * a.W * a.Cin + g * a.cgs;
        for (int i0 = tid; i0 < nitems; i0 += NT * UNR) {
          float4 va[UNR], vb[UNR];
          bool ok[UNR];
          int dst[UNR];
          int emit[UNR];      // xb_out: element offset (inside the sample's channel group) of an item this tile owns, else -1
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * NT;
            const int ic = min(i, nitems - 1);
            const int p = ic >> a.log2nq, q = ic & (nq - 1);
            const int hy = (int)(((unsigned)p * (unsigned)a.wt_magic) >> 24);
            const int hx = p - hy * a.WT;
            int cy = iy0 + hy, cx = ix0 + hx;
            bool v = cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
            v = v && (((cy | cx) & dsh) == 0);  // zero-stuffed operand: odd positions are zeros
            cy >>= dsh; cx >>= dsh;
            ok[u] = v;
            dst[u] = (i < nitems) ? q * a.NPIXP + p : dummy;
            const size_t eo = ((size_t)(v ? cy : 0) * a.W + (v ? cx : 0)) * a.Cin + q * 8;
            if (EMIT && !PRECISE) {
              const bool own = emit_xb && v && i < nitems && cy >= own_y0 && cy < own_y1 && cx >= own_x0 && cx < own_x1;
              emit[u] = own ? (int)eo : -1;
            }
            if (!PRECISE && a.x_bf16) {   // workgroup-uniform: a final bf16 activation is copied, not converted
              va[u] = __builtin_bit_cast(float4, *reinterpret_cast<const uint4*>(
                                                     reinterpret_cast<const unsigned short*>(a.x) + (size_t)bx * a.H * a.W * a.Cin + g * a.cgs + eo));
            } else {
              va[u] = *reinterpret_cast<const float4*>(xb + eo);
              vb[u] = *reinterpret_cast<const float4*>(xb + eo + 4);
            }
          }
          if (!PRECISE && a.x_bf16) {
            if (xf_identity) {     // a final activation: copied
#pragma unroll
              for (int u = 0; u < UNR; ++u) sAhi[dst[u]] = ok[u] ? __builtin_bit_cast(uint4, va[u]) : uint4{0, 0, 0, 0};
            } else {               // a raw conv output stored as bf16 (hdrsky_conv_desc.y_bf16 in front of a norm layer): the
                                   // producer's normalisation + activation on the widened values, as for fp32 storage
#pragma unroll
              for (int u = 0; u < UNR; ++u) {
                const uint4 b8 = __builtin_bit_cast(uint4, va[u]);
                const unsigned wds[4] = {b8.x, b8.y, b8.z, b8.w};
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const float in = __builtin_bit_cast(float, (j & 1) ? (wds[j >> 1] & 0xffff0000u) : (wds[j >> 1] << 16));
                  v[j] = ok[u] ? leaky(in * sc8[j] + sh8[j], slope) : 0.f;
                }
                uint4 hi, lo;
                pack8<false>(v, hi, lo);
                sAhi[dst[u]] = hi;
                if (EMIT && emit_xb && emit[u] >= 0) *reinterpret_cast<uint4*>(xbo + emit[u]) = hi;
              }
            }
            continue;
          }
#pragma unroll
          for (int u = 0; u < UNR; ++u) {
            const float in[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
            float v[8];
            if (xf_identity) {   // workgroup-uniform: no producer transform (already normalised / activated input)
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = ok[u] ? in[j] : 0.f;
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float t = leaky(in[j] * sc8[j] + sh8[j], slope);
                v[j] = ok[u] ? t : 0.f;
              }
            }
            uint4 hi, lo;
            pack8<PRECISE>(v, hi, lo);
            sAhi[dst[u]] = hi;
            if (PRECISE) sAlo[dst[u]] = lo;
            if (EMIT && !PRECISE && emit_xb && emit[u] >= 0) *reinterpret_cast<uint4*>(xbo + emit[u]) = hi;
          }
        }
      }
    }
    if (g == 0) { HDRSKY_STAMP(2) }

    if (DB) {
      // ---- direct-B main loop: no LDS for weights, no barriers.  Each wave streams ITS B fragments
      // (16-byte rows of the packed image, L2/L1-resident) into a rotating window of DPF k-steps of
      // registers; a fragment's register is refilled right after the MFMAs that consumed it issue.
      __syncthreads();  // operand planes staged
      if constexpr (!EARLY_B) fill_b(g);
      if constexpr (!NARROW) {
        // Table-driven form (see the prologue): per group of DPF k-steps one LDS read fetches the A offsets of the next
        // steps and the filter refill offsets, v_readlane turns them into scalars - no tap / channel-block arithmetic,
        // no 64-bit index products in the loop (the generic form below spent ~30 scalar instructions per k-step on
        // them, against two MFMAs; the scalar unit serves a SIMD every fourth cycle).  A fragments are read one k-step
        // ahead into the register set of the step's parity.
        const uint4* wgh = wlh + (size_t)(g << a.log2cbg) * kstride;
        const uint4* wgl = PRECISE ? wll + (size_t)(g << a.log2cbg) * kstride : nullptr;
        uint4 ah[2][MI], al[2][MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {   // step 0: tap 0, channel block 0 = offset 0
          ah[0][mi] = *reinterpret_cast<const uint4*>(smem + abase[mi]);
          if (PRECISE) al[0][mi] = *reinterpret_cast<const uint4*>(smem + a.off_alo + abase[mi]);
        }
        static_assert((DPF & 1) == 0, "the register set of a step is chosen by its parity");
        auto group = [&](int ks0, auto checked) {
          const int2 tv = sKtab[ks0 + (lane & (DPF - 1))];
          const int rem = a.ksg - ks0;
#pragma unroll
          for (int j = 0; j < DPF; ++j) {
            if (!decltype(checked)::value || j < rem) {
              const int aoff = __builtin_amdgcn_readlane(tv.x, j);
              const unsigned o = (unsigned)__builtin_amdgcn_readlane(tv.y, j);
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) {
                ah[(j + 1) & 1][mi] = *reinterpret_cast<const uint4*>(smem + abase[mi] + aoff);
                if (PRECISE) al[(j + 1) & 1][mi] = *reinterpret_cast<const uint4*>(smem + a.off_alo + abase[mi] + aoff);
              }
#pragma unroll
              for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                  if (PRECISE) {
                    acc[mi][ni] = mfma16(al[j & 1][mi], bqh[j][ni], acc[mi][ni]);
                    acc[mi][ni] = mfma16(ah[j & 1][mi], bql[j][ni], acc[mi][ni]);
                  }
                  acc[mi][ni] = DIRECT ? mfma16(bqh[j][ni], ah[j & 1][mi], acc[mi][ni]) : mfma16(ah[j & 1][mi], bqh[j][ni], acc[mi][ni]);
                }
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) {   // refill with step ks+DPF (clamped in the table: surplus loads are unused)
                bqh[j][ni] = wgh[o + ni * 16];
                if (PRECISE) bql[j][ni] = wgl[o + ni * 16];
              }
            }
          }
        };
        int ks0 = 0;
        for (; ks0 + DPF <= a.ksg; ks0 += DPF) group(ks0, std::false_type{});
        if (ks0 < a.ksg) group(ks0, std::true_type{});
      } else {
        // A fragments are read from LDS one k-step ahead (two register sets, selected by the unrolled step's parity), so
        // the read latency overlaps the MFMAs of the current step instead of preceding them.
        auto aoff_of = [&](int ks) {
          if (NARROW) {
            const int tap = min(ks * 4 + kq, a.ntaps - 1);
            const int ky = (tap * a.kw_magic) >> 16;
            return (ky * a.WT + (tap - ky * a.KW)) * 16;
          }
          const int tap = ks >> a.log2cbg, cb = ks & ((1 << a.log2cbg) - 1);
          const int ky = (tap * a.kw_magic) >> 16;
          return (ky * a.WT + (tap - ky * a.KW) + cb * 4 * a.NPIXP) * 16;
        };
        uint4 ah[2][MI], al[2][MI];
        {
          const int aoff = aoff_of(0);
  #pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            ah[0][mi] = *reinterpret_cast<const uint4*>(smem + abase[mi] + aoff);
            if (PRECISE) al[0][mi] = *reinterpret_cast<const uint4*>(smem + a.off_alo + abase[mi] + aoff);
          }
        }
        static_assert((DPF & 1) == 0, "the register set of a step is chosen by its parity");
        for (int ks0 = 0; ks0 < a.ksg; ks0 += DPF) {
          const bool full = ks0 + DPF <= a.ksg;
  #pragma unroll
          for (int j = 0; j < DPF; ++j) {
            const int ks = ks0 + j;
            if (full || ks < a.ksg) {
              if (ks + 1 < a.ksg) {
                const int aoff = aoff_of(ks + 1);
  #pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                  ah[(j + 1) & 1][mi] = *reinterpret_cast<const uint4*>(smem + abase[mi] + aoff);
                  if (PRECISE) al[(j + 1) & 1][mi] = *reinterpret_cast<const uint4*>(smem + a.off_alo + abase[mi] + aoff);
                }
              }
  #pragma unroll
              for (int mi = 0; mi < MI; ++mi)
  #pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                  if (PRECISE) {
                    acc[mi][ni] = mfma16(al[j & 1][mi], bqh[j][ni], acc[mi][ni]);
                    acc[mi][ni] = mfma16(ah[j & 1][mi], bql[j][ni], acc[mi][ni]);
                  }
                  acc[mi][ni] = DIRECT ? mfma16(bqh[j][ni], ah[j & 1][mi], acc[mi][ni]) : mfma16(ah[j & 1][mi], bqh[j][ni], acc[mi][ni]);
                }
              const unsigned o = (unsigned)kp_of(g, ks + DPF) * kstride;  // clamped at the end: surplus loads are unused
  #pragma unroll
              for (int ni = 0; ni < NI; ++ni) {
                bqh[j][ni] = wlh[o + ni * 16];
                if (PRECISE) bql[j][ni] = wll[o + ni * 16];
              }
            }
          }
        }
      }
      // all waves done with the operand planes before the next group restages them / the epilogue tile overwrites
      // them; not needed after the last group when the epilogue tile has its own LDS (off_out != 0)
      if (g + 1 < a.ngroups || (!DIRECT && a.off_out == 0)) __syncthreads();
    } else {
      // ---- B ring: LDS double buffer; chunk ch+1 is copied by LDS-DMA into the other buffer (free since the barrier
      // that ended iteration ch-1) while the MFMAs of chunk ch run: no registers, no ds_write, nothing in scratch.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // chunk 0 (DMA) landed
      __syncthreads();
      unsigned long long tl = 0, tc = 0, ts = 0, tb = 0;  // debug: cycles in load-issue / MFMA / LDS-store(+load wait) / barrier
      for (int ch = 0; ch < nchunks; ++ch) {
        const bool dbg = a.stamps != nullptr;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (dbg) t0 = __builtin_amdgcn_s_memtime();
        HDRSKY_DMA_B(ch + 1, (ch + 1) & 1)
        if (dbg) t1 = __builtin_amdgcn_s_memtime();
        if (ch + 1 < nchunks)
          compute_chunk<MI, NI, BN, KC, BPLANES, BITEMS, NARROW, PRECISE, false>(a, smem, sB, sTap, ch, kq, abase, bbase, acc);
        else
          compute_chunk<MI, NI, BN, KC, BPLANES, BITEMS, NARROW, PRECISE, true>(a, smem, sB, sTap, ch, kq, abase, bbase, acc);
        if (dbg) { asm volatile("" ::"v"(acc[0][0][0])); t2 = __builtin_amdgcn_s_memtime(); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // chunk ch+1 landed
        if (dbg) t3 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (dbg) { t4 = __builtin_amdgcn_s_memtime(); tl += t1 - t0; tc += t2 - t1; ts += t3 - t2; tb += t4 - t3; }
      }
      if (a.stamps != nullptr && tid == 0 && g == 0) {
        unsigned long long* d = a.stamps + (size_t)gridDim.x * 8 + (size_t)blockIdx.x * 4;
        d[0] = tl; d[1] = tc; d[2] = ts; d[3] = tb;
      }
    }
  }

  HDRSKY_STAMP(3)
  if constexpr (DIRECT) {
    // ---- epilogue from the registers: lane (kq, lr) holds channels n0 + (wn NI + ni) 16 + 4 kq .. + 3 of the pixels lr of its MI fragments
    float cs[NI][4], cq[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) { cs[ni][e] = 0.f; cq[ni][e] = 0.f; }
    const bool vec4 = (a.Cout & 3) == 0;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + (wn * NI + ni) * 16 + kq * 4;
      float bias4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bias4[e] = (bias != nullptr && n + e < a.Cout) ? bias[n + e] : 0.f;
      const bool vec = vec4 && (n + 3 < a.Cout);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int f = wm * MI + mi;
        int oy = oy0 + f / FPR, ox = ox0 + (f % FPR) * 16 + lr;
        if (PH) { oy = 2 * oy + (ph >> 1); ox = 2 * ox + (ph & 1); }
        if (oy < a.Ho && ox < a.Wo && n < a.Cout) {
          float v[4] = {acc[mi][ni][0] + bias4[0], acc[mi][ni][1] + bias4[1], acc[mi][ni][2] + bias4[2], acc[mi][ni][3] + bias4[3]};
          const size_t idx = ((size_t)(b * a.Ho + oy) * a.Wo + ox) * a.Cout + n;
          const size_t ridx = ((size_t)(br * a.Ho + oy) * a.Wo + ox) * a.Cout + n;      // inside the group's residual / mask tensor
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e < a.Cout) { cs[ni][e] += v[e]; cq[ni][e] += v[e] * v[e]; }
            v[e] = leaky(v[e], a.out_slope);
          }
          if (vec) {
            if (residual != nullptr) {
              if (a.res_mode == 1) {   // bf16 activated tensor: gradient mask of the activation behind this data gradient
                const uint2 mk = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(residual) + ridx);
                const float m4[4] = {__builtin_bit_cast(float, mk.x << 16), __builtin_bit_cast(float, mk.x & 0xffff0000u),
                                     __builtin_bit_cast(float, mk.y << 16), __builtin_bit_cast(float, mk.y & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (m4[e] > 0.f ? 1.f : a.mask_slope);
              } else {
                const float4 r = *reinterpret_cast<const float4*>(residual + ridx);
                v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
              }
            }
            if (a.final_relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (a.y_bf16)
              *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.y) + idx) =
                  uint2{(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
            else
              *reinterpret_cast<float4*>(a.y + idx) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < a.Cout) {
                float o = v[e];
                if (residual != nullptr) o += residual[ridx + e];
                if (a.final_relu) o = fmaxf(o, 0.f);
                a.y[idx + e] = o;
              }
          }
        }
      }
    }
    if (a.want_stats) {
      // sum over the 16 pixels (lanes lr) of a k-quarter row, then over the waves that share the column block
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { cs[ni][e] += __shfl_xor(cs[ni][e], o); cq[ni][e] += __shfl_xor(cq[ni][e], o); }
        }
      if (lr == 0) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int col = (wn * NI + ni) * 16 + kq * 4 + e;
            sStat[(wm * BN + col) * 2 + 0] = cs[ni][e];
            sStat[(wm * BN + col) * 2 + 1] = cq[ni][e];
          }
      }
      __syncthreads();
      if (tid < BN && n0 + tid < a.Cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          s += sStat[(w * BN + tid) * 2 + 0];
          q += sStat[(w * BN + tid) * 2 + 1];
        }
        const int nparts = a.tiles_x * a.tiles_y;
        float* dst = a.stats + ((size_t)(b * nparts + ty * a.tiles_x + tx) * 2) * a.Cout + n0 + tid;
        dst[0] = s;
        dst[a.Cout] = q;
      }
    }
    HDRSKY_STAMP(5)
    return;
  }
  // ---- epilogue: accumulators -> LDS tile [BM][BN] -> coalesced 16-byte rows ------------------------
  // (the last __syncthreads of the ring guarantees every wave is done reading the operand planes)
  float* sOut = reinterpret_cast<float*>(smem + a.off_out);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        sOut[((wm * MI + mi) * 16 + kq * 4 + j) * BNP + (wn * NI + ni) * 16 + lr] = acc[mi][ni][j];
  __syncthreads();
  constexpr int C4 = BN / 4;          // float4 columns
  constexpr int PPI = NT / C4;        // pixels per pass
  const int c4 = tid % C4;
  const int n = n0 + c4 * 4;
  float bias4[4], cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) bias4[e] = (bias != nullptr && n + e < a.Cout) ? bias[n + e] : 0.f;
  const bool vec = ((a.Cout & 3) == 0) && (n + 3 < a.Cout);
#pragma unroll
  for (int it = 0; it * PPI < BM; ++it) {
    const int m = it * PPI + tid / C4;           // tile-local pixel (fragment-major order)
    const int f = m >> 4;
    int oy = oy0 + f / FPR, ox = ox0 + (f % FPR) * 16 + (m & 15);
    if (PH) { oy = 2 * oy + (ph >> 1); ox = 2 * ox + (ph & 1); }
    if (m < BM && oy < a.Ho && ox < a.Wo) {
      const float4 t = *reinterpret_cast<const float4*>(sOut + m * BNP + c4 * 4);
      float v[4] = {t.x + bias4[0], t.y + bias4[1], t.z + bias4[2], t.w + bias4[3]};
      const size_t idx = ((size_t)(b * a.Ho + oy) * a.Wo + ox) * a.Cout + n;
      const size_t ridx = ((size_t)(br * a.Ho + oy) * a.Wo + ox) * a.Cout + n;      // inside the group's residual / mask tensor
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < a.Cout) { cs[e] += v[e]; cq[e] += v[e] * v[e]; }
        v[e] = leaky(v[e], a.out_slope);
      }
      if (vec) {
        if (residual != nullptr) {
          if (a.res_mode == 1) {   // bf16 activated tensor: gradient mask of the activation behind this data gradient
            const uint2 mk = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(residual) + ridx);
            const float m4[4] = {__builtin_bit_cast(float, mk.x << 16), __builtin_bit_cast(float, mk.x & 0xffff0000u),
                                 __builtin_bit_cast(float, mk.y << 16), __builtin_bit_cast(float, mk.y & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= (m4[e] > 0.f ? 1.f : a.mask_slope);
          } else {
            const float4 r = *reinterpret_cast<const float4*>(residual + ridx);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
          }
        }
        if (a.final_relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.y_bf16)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.y) + idx) =
              uint2{(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
        else
          *reinterpret_cast<float4*>(a.y + idx) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < a.Cout) {
            float o = v[e];
            if (residual != nullptr) o += residual[ridx + e];
            if (a.final_relu) o = fmaxf(o, 0.f);
            a.y[idx + e] = o;
          }
      }
    }
  }
  if (a.want_stats) {
    // lanes holding the same float4 column are C4 apart inside a wave
#pragma unroll
    for (int o = C4; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { cs[e] += __shfl_xor(cs[e], o); cq[e] += __shfl_xor(cq[e], o); }
    }
    if (lane < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sStat[(wave * BN + c4 * 4 + e) * 2 + 0] = cs[e];
        sStat[(wave * BN + c4 * 4 + e) * 2 + 1] = cq[e];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        s += sStat[(w * BN + tid) * 2 + 0];
        q += sStat[(w * BN + tid) * 2 + 1];
      }
      const int nparts = a.tiles_x * a.tiles_y;
      float* dst = a.stats + ((size_t)(b * nparts + ty * a.tiles_x + tx) * 2) * a.Cout + n0 + tid;
      dst[0] = s;
      dst[a.Cout] = q;
    }
  }
  HDRSKY_STAMP(5)
}

// ---------------------------------------------------------------------------------------------------
// weight packing: fp32 HWIO -> [kstep][4][Npad][8] bf16 (hi / lo planes)
// ---------------------------------------------------------------------------------------------------
// one 16-byte group of the packed image (8 consecutive k of one column; group index g = element index / 8) from the fp32
// HWIO filter.  A thread per group: the index arithmetic (five divisions) once per 8 elements, the eight source values of
// adjacent lanes (adjacent columns n) are adjacent in memory, one 16-byte store per plane.  (One element per thread with
// 2-byte stores took 48 us for the step's 16 M elements, all of it on the serial tail of the step.)
__device__ __forceinline__ void pack_group(const float* __restrict__ w, int KH, int KW, int Cin, int Cout, int Npad,
                                           int narrow, int flip, int ksteps, size_t g, unsigned short* __restrict__ hi,
                                           unsigned short* __restrict__ lo) {
  // thread -> group: columns n fastest for the plain image (the 8 source values of adjacent lanes are adjacent in memory);
  // for the transpose_flip image the source is contiguous along c, so there the four 8-channel quarters q of a k-step
  // are the fastest index (four lanes read one 128-byte line) and the destination index is permuted instead
  int n, q, kp;
  if (flip) {
    const size_t per = (size_t)4 * Npad;
    kp = (int)(g / per);
    const int gl = (int)(g - (size_t)kp * per);
    q = gl & 3; n = gl >> 2;
    g = ((size_t)kp * 4 + q) * Npad + n;
  } else {
    size_t r = g;
    n = r % Npad; r /= Npad;
    q = r & 3;
    kp = (int)(r >> 2);
  }
  int tap, c0;
  if (narrow) { tap = kp * 4 + q; c0 = 0; }
  else { const int cin32 = Cin >> 5; tap = kp / cin32; c0 = (kp % cin32) * 32 + q * 8; }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (kp < ksteps && tap < KH * KW && n < Cout) {
    int ky = tap / KW, kx = tap % KW;
    if (flip) {
      // packed filter w'[ky,kx,c(=co of w),n(=ci of w)] = w[KH-1-ky, KW-1-kx, n, c]; w is [KH,KW,Cout',Cin']
      ky = KH - 1 - ky; kx = KW - 1 - kx;
      const float* src = w + ((size_t)(ky * KW + kx) * Cout + n) * Cin + c0;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (c0 + j < Cin) v[j] = src[j];
    } else {
      const float* src = w + ((size_t)(ky * KW + kx) * Cin + c0) * Cout + n;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (c0 + j < Cin) v[j] = src[(size_t)j * Cout];
    }
  }
  uint4 h8, l8;
  if (lo != nullptr) pack8<true>(v, h8, l8); else pack8<false>(v, h8, l8);
  reinterpret_cast<uint4*>(hi)[g] = h8;
  if (lo != nullptr) reinterpret_cast<uint4*>(lo)[g] = l8;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, int KH, int KW, int Cin, int Cout, int Npad,
                                    int narrow, int flip, int ksteps, unsigned short* __restrict__ hi,
                                    unsigned short* __restrict__ lo) {
  const size_t groups = (size_t)(ksteps + 1) * 4 * Npad;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x)
    pack_group(w, KH, KW, Cin, Cout, Npad, narrow, flip, ksteps, g, hi, lo);
}

// Multi-tensor re-pack after an optimizer step: one launch for every conv filter of a network.
// jobs[j] = {w, hi, lo, KH, KW, Cin, Cout, flip, first_block} (9 x int64); a block packs 2048 elements = 256 groups.
__global__ void __launch_bounds__(256) pack_multi_kernel(const long long* __restrict__ jobs, int njobs) {
  int lo_j = 0, hi_j = njobs - 1;
  while (lo_j < hi_j) {  // last job whose first_block <= blockIdx.x
    const int mid = (lo_j + hi_j + 1) >> 1;
    if (jobs[mid * 9 + 8] <= (long long)blockIdx.x) lo_j = mid; else hi_j = mid - 1;
  }
  const long long* jb = jobs + lo_j * 9;
  const float* w = reinterpret_cast<const float*>(jb[0]);
  unsigned short* hi = reinterpret_cast<unsigned short*>(jb[1]);
  unsigned short* lo = reinterpret_cast<unsigned short*>(jb[2]);
  const int KH = (int)jb[3], KW = (int)jb[4], Cin = (int)jb[5], Cout = (int)jb[6], flip = (int)jb[7];
  const int narrow = Cin <= 8 ? 1 : 0;
  const int ksteps = narrow ? (KH * KW + 3) / 4 : KH * KW * (Cin / 32);
  const int Npad = (Cout + 63) / 64 * 64;
  const size_t groups = (size_t)(ksteps + 1) * 4 * Npad;
  const size_t g = (size_t)(blockIdx.x - (int)jb[8]) * 256 + threadIdx.x;
  if (g < groups) pack_group(w, KH, KW, Cin, Cout, Npad, narrow, flip, ksteps, g, hi, lo);
}

int conv_ksteps(int KH, int KW, int Cin) {
  if (Cin <= 8) return cdiv(KH * KW, 4);
  return KH * KW * (Cin / 32);
}

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

struct TileCfg { int wm, wn, mi, ni, tw, db; };

constexpr int HDRSKY_EPHASE_FALLBACK = -1000;   // internal: launch_conv declines the phase form of a stride-2 data gradient

template <int WM, int WN, int MI, int NI, int TW, bool NARROW, bool PRECISE, bool DB, bool PH, bool EMIT = false, bool PAIR = false, bool UP = false>
int launch_conv(ConvKArgs& a, hipStream_t stream) {
  constexpr int BM = WM * MI * 16, BN = WN * NI * 16, TH = BM / TW;
  a.tiles_x = cdiv(PH ? cdiv(a.Wo, 2) : a.Wo, TW);    // phase mode: tiles of one phase's grid, four phases per sample
  a.tiles_y = cdiv(PH ? cdiv(a.Ho, 2) : a.Ho, TH);
  a.nblocks = cdiv(a.Cout, BN);
  a.HT = (TH - 1) * a.stride + a.KH;
  a.WT = (TW - 1) * a.stride + a.KW;
  a.NPIX = a.HT * a.WT;
  a.NPIXP = roundup(a.NPIX, 16) + 1;  // = 1 (mod 16): conflict-free ds_write_b128 of the staging, <=1 two-way slot on reads
  a.wt_magic = ((1 << 24) + a.WT - 1) / a.WT;
  a.kw_magic = (65536 + a.KW - 1) / a.KW;
  a.ntaps = a.KH * a.KW;
  const int bplanes = PRECISE ? 2 : 1;
  constexpr int KC = kc_for(BN);
  const int b_bytes = DB ? 0 : 2 * bplanes * KC * 4 * BN * 16;
  constexpr int NW = WM * WN;
  const int misc = 2 * a.Cin * 4 + roundup(a.ntaps, 4) * 4 + NW * BN * 2 * 4;
  const int budget = 160 * 1024 - b_bytes - roundup(misc, 16) - 64;
  if (NARROW) {
    a.cgs = 8; a.ngroups = 1; a.log2nq = 0; a.log2cbg = 0;
    a.ksg = cdiv(a.ntaps, 4);
  } else {
    // channel groups are powers of two (index arithmetic by shifts) that divide Cin: all of Cin when it is one, else its
    // largest power-of-two divisor (1152 = 9 x 128: the 1x1 conv over the k*k*C gathered channels of a distortion-aware layer)
    int cgs = (a.Cin & (a.Cin - 1)) ? (a.Cin & -a.Cin) : a.Cin;
    while (cgs > 32 && (cgs / 8) * a.NPIXP * 16 * bplanes > budget) cgs >>= 1;
    if ((cgs / 8) * a.NPIXP * 16 * bplanes > budget || (a.Cin % cgs) != 0) return HDRSKY_EUNSUPPORTED;
    a.cgs = cgs; a.ngroups = a.Cin / cgs; a.log2nq = ilog2(cgs / 8); a.log2cbg = ilog2(cgs / 32);
    if ((1 << a.log2nq) != cgs / 8) return HDRSKY_EUNSUPPORTED;
    a.ksg = a.ntaps * (cgs / 32);
  }
  // an odd filter's phases pad their tap grid with the packed image's single all-zero k-step, which the direct-B table
  // addresses relative to a channel group's base: with several groups the zero-stuffed form runs instead
  if (PH && ((a.KHf | a.KWf) & 1) && a.ngroups > 1) return HDRSKY_EPHASE_FALLBACK;
  const int a_plane = (NARROW ? 1 : a.cgs / 8) * a.NPIXP * 16;
  a.off_alo = a_plane;
  a.off_b = a_plane * bplanes;
  a.off_ss = a.off_b + b_bytes;
  a.off_tap = a.off_ss + 2 * a.Cin * 4;
  a.off_stat = roundup(a.off_tap + a.ntaps * 4, 16);
  int lds = a.off_stat + NW * BN * 2 * 4;
  // the epilogue re-uses the operand planes (from offset 0) as a [BM][BN+4] fp32 tile; it must not reach
  // the stat scratch
  constexpr bool DIRECT = DB && !PRECISE && (HDRSKY_EPI_DIRECT != 0);     // epilogue from the registers: no LDS tile
  const int out_bytes = DIRECT ? 0 : BM * (BN + 4) * 4;
  if (out_bytes > a.off_stat) { a.off_stat = roundup(out_bytes, 16); lds = a.off_stat + NW * BN * 2 * 4; }
  a.off_ktab = roundup(lds, 16);
  if (DB && !NARROW) lds = a.off_ktab + (a.ksg + 8) * 8;   // k-step table of the direct-B loop
  // direct-B variant: the epilogue tile may get LDS of its own, so that a wave that has finished its main loop can write its
  // accumulators without waiting for the slowest wave to stop reading the operand planes (rounds 3-4, threshold 80 KB; since
  // round 5 the threshold is 0 = always aliased: hooks.h, HDRSKY_CONV_EPI_LDS)
  a.off_out = 0;
  if (DB && roundup(lds, 16) + out_bytes <= hdrsky_hooks().conv_epi_lds * 1024) { a.off_out = roundup(lds, 16); lds = a.off_out + out_bytes; }   // (two workgroups per CU must still fit)
  if (lds > 160 * 1024) return HDRSKY_EUNSUPPORTED;
  if ((a.upsample == 2) != UP) return HDRSKY_EUNSUPPORTED;     // (UP: the instantiations with the resize-fused staging, dispatch_tile_up)
  auto kern = conv_igemm_kernel<WM, WN, MI, NI, TW, NARROW, PRECISE, DB, PH, EMIT, PAIR, UP>;
  static std::atomic<int> max_lds_set{0};
  if (lds > max_lds_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return HDRSKY_ELAUNCH;
    max_lds_set = 160 * 1024;
  }
  const int grid = a.B * (PH ? 4 : 1) * a.tiles_y * a.tiles_x * a.nblocks;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, a);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

template <bool NARROW, bool PRECISE, bool PH = false, bool EMIT = false>
int dispatch_tile(ConvKArgs& a, const TileCfg& t, hipStream_t s) {
#define HDRSKY_CASE(WM_, WN_, MI_, NI_, TW_)                                              \
  if (!t.db && t.wm == WM_ && t.wn == WN_ && t.mi == MI_ && t.ni == NI_ && t.tw == TW_) \
    return launch_conv<WM_, WN_, MI_, NI_, TW_, NARROW, PRECISE, false, PH, EMIT>(a, s);
#define HDRSKY_CASE_DB(WM_, WN_, MI_, NI_, TW_)                                           \
  if (t.db && t.wm == WM_ && t.wn == WN_ && t.mi == MI_ && t.ni == NI_ && t.tw == TW_)  \
    return launch_conv<WM_, WN_, MI_, NI_, TW_, NARROW, PRECISE, true, PH, EMIT>(a, s);
#ifdef HDRSKY_ONE_TILE      // (diagnostic builds: one instantiation per mode, seconds instead of minutes per compile)
  HDRSKY_CASE_DB(2, 4, 4, 1, 32)
#else
  // LDS-ring variant
  HDRSKY_CASE(2, 2, 4, 2, 32) HDRSKY_CASE(2, 2, 2, 2, 32) HDRSKY_CASE(2, 2, 2, 2, 16)
  HDRSKY_CASE(4, 1, 4, 2, 32) HDRSKY_CASE(4, 1, 2, 2, 32) HDRSKY_CASE(2, 2, 2, 1, 32) HDRSKY_CASE(2, 2, 2, 1, 16)
  HDRSKY_CASE(4, 1, 4, 1, 32) HDRSKY_CASE(4, 1, 2, 1, 32) HDRSKY_CASE(4, 1, 1, 1, 16)
  HDRSKY_CASE(2, 4, 2, 1, 32) HDRSKY_CASE(4, 2, 2, 2, 32) HDRSKY_CASE(4, 2, 2, 1, 32) HDRSKY_CASE(8, 1, 2, 2, 32)
  HDRSKY_CASE(8, 1, 4, 2, 32) HDRSKY_CASE(2, 4, 2, 1, 16) HDRSKY_CASE(8, 1, 2, 1, 32)
  // direct-B variant (weights streamed per wave, barrier-free main loop)
  HDRSKY_CASE_DB(1, 4, 4, 1, 32) HDRSKY_CASE_DB(2, 4, 4, 1, 32) HDRSKY_CASE_DB(1, 8, 4, 1, 32) HDRSKY_CASE_DB(2, 2, 4, 1, 32)
  HDRSKY_CASE_DB(4, 2, 4, 1, 32) HDRSKY_CASE_DB(2, 4, 2, 1, 32) HDRSKY_CASE_DB(4, 1, 4, 1, 32) HDRSKY_CASE_DB(8, 1, 4, 1, 32)
  HDRSKY_CASE_DB(1, 4, 4, 1, 16) HDRSKY_CASE_DB(1, 4, 2, 1, 16) HDRSKY_CASE_DB(2, 2, 4, 2, 32) HDRSKY_CASE_DB(2, 4, 4, 2, 32)
  HDRSKY_CASE_DB(1, 8, 2, 1, 16)
  // round 5: 64 px x 64 ch per WAVE (MI = NI = 4: an A fragment read from LDS feeds four MFMAs - at NI = 1 the kernel is bound by its
  // LDS fragment reads, 1 KB per MFMA against 0.5 KB per MFMA-time of LDS bandwidth), four waves with the full register budget
  HDRSKY_CASE_DB(4, 1, 4, 4, 32) HDRSKY_CASE_DB(2, 2, 4, 4, 32) HDRSKY_CASE_DB(4, 1, 2, 4, 32) HDRSKY_CASE_DB(1, 4, 4, 4, 32)
#endif
#undef HDRSKY_CASE_DB
#undef HDRSKY_CASE
  return HDRSKY_EUNSUPPORTED;
}

// The resize-fused staging's instantiations (hdrsky_conv_desc.upsample == 2: engine.decode's four deconvolutions and the
// eager / fp32-class steps): the tiles the table gives those layers at 32x128 and 128x512; another tile falls back to the
// class's round-1 entry.
template <bool PRECISE>
int dispatch_tile_up(ConvKArgs& a, TileCfg t, hipStream_t s) {
  for (int attempt = 0; attempt < 2; ++attempt) {
#define HDRSKY_UCASE(WM_, WN_, MI_, NI_, TW_, DB_)                                                         \
    if ((t.db != 0) == DB_ && t.wm == WM_ && t.wn == WN_ && t.mi == MI_ && t.ni == NI_ && t.tw == TW_)    \
      return launch_conv<WM_, WN_, MI_, NI_, TW_, false, PRECISE, DB_, false, false, false, true>(a, s);
    HDRSKY_UCASE(2, 4, 4, 1, 32, true) HDRSKY_UCASE(4, 2, 4, 1, 32, true)
#ifndef HDRSKY_ONE_TILE
    HDRSKY_UCASE(2, 2, 4, 2, 32, true) HDRSKY_UCASE(2, 4, 4, 2, 32, true)
    HDRSKY_UCASE(1, 4, 4, 1, 32, true) HDRSKY_UCASE(2, 2, 4, 1, 32, true) HDRSKY_UCASE(1, 8, 4, 1, 32, true) HDRSKY_UCASE(2, 4, 2, 1, 32, true)
    HDRSKY_UCASE(8, 1, 4, 2, 32, false) HDRSKY_UCASE(2, 2, 2, 2, 32, false) HDRSKY_UCASE(2, 2, 4, 2, 32, false) HDRSKY_UCASE(4, 1, 4, 2, 32, false)
#endif
#undef HDRSKY_UCASE
    t = a.Cout >= 64 ? TileCfg{2, 4, 4, 1, 32, 1} : TileCfg{4, 2, 4, 1, 32, 1};
  }
  return HDRSKY_EUNSUPPORTED;
}

// The paired launches' instantiations (single-product mode, no phases, no emit): the tiles the decoder layers and their data
// gradients take at batch 32 and at the 128x512 network's batch 8; another tile -> HDRSKY_EUNSUPPORTED (the caller issues the
// two launches)
template <bool NARROW>
int dispatch_tile_pair(ConvKArgs& a, const TileCfg& t, hipStream_t s) {
#define HDRSKY_PCASE(WM_, WN_, MI_, NI_, TW_, DB_)                                                         \
  if ((t.db != 0) == DB_ && t.wm == WM_ && t.wn == WN_ && t.mi == MI_ && t.ni == NI_ && t.tw == TW_)    \
    return launch_conv<WM_, WN_, MI_, NI_, TW_, NARROW, false, DB_, false, false, true>(a, s);
#ifndef HDRSKY_ONE_TILE
  HDRSKY_PCASE(2, 4, 4, 1, 32, true) HDRSKY_PCASE(4, 2, 4, 1, 32, true) HDRSKY_PCASE(8, 1, 4, 1, 32, true) HDRSKY_PCASE(4, 1, 4, 1, 32, true)
  HDRSKY_PCASE(2, 4, 4, 2, 32, true) HDRSKY_PCASE(2, 2, 4, 1, 32, true) HDRSKY_PCASE(8, 1, 4, 2, 32, false) HDRSKY_PCASE(2, 4, 2, 1, 32, true)
  HDRSKY_PCASE(2, 2, 4, 2, 32, true) HDRSKY_PCASE(4, 1, 4, 4, 32, true)      // (the 128x512 network's decoder entries)
#endif
#undef HDRSKY_PCASE
  return HDRSKY_EUNSUPPORTED;
}

// Tile heuristic: widest N block the layer fills, then the largest pixel tile that still
// yields >= ~1 workgroup per CU (256 CUs), preferring more workgroups for small problems.
TileCfg choose_tile_r4(const hdrsky_conv_desc* d);

// Round 5: the table re-measured in a second regime - the chip full of waves of the launch's own kind (three streams replaying
// graphs of the launch: profiles/tile_sweep.py -> profiles/r05_tile_sweep.txt) - beside the launch's latency alone on the chip,
// which rounds 1-3 tuned (choose_tile_r4 below).  Saturated, the kernel is bound by its LDS fragment reads (one 1 KB A fragment
// per MFMA with NI = 1: twice the MFMA time), the staging burst and the epilogue burst.  A first version that took the best
// SATURATED tile of every class (-12 % summed over the step's launches in that regime) made the step 0.8 % SLOWER
// (profiles/r05_tile_table_ab.txt): the step's dependent chains pay the latency of a launch, and a fatter tile holds its compute
// unit longer against the launches of the other two streams.  What is kept are the entries that win saturated by >= 10 % WITHOUT
// losing alone.  `ph`: the launch is a stride-2 data gradient by output phases.
// HDRSKY_TILE_TABLE=4: round 4's table (switch: A/B and bit-identity tests - a tile changes the statistics partials' order).
TileCfg choose_tile(const hdrsky_conv_desc* d, bool ph = false) {
  const HdrskyHooks& hk = hdrsky_hooks();
  TileCfg t = choose_tile_r4(d);
  if (hk.tile_table == 4 || hk.tile.set || d->compute == HDRSKY_BF16X3) return t;
  const long M = (long)d->B * d->Ho * d->Wo;
  for (int i = 0; i < hk.ntile_rules; ++i) {       // HDRSKY_TILE_RULES (A/B runs)
    const HdrskyTileRule& r = hk.tile_rules[i];
    if (d->Cout >= r.lo[0] && d->Cout <= r.hi[0] && M >= r.lo[1] && M <= r.hi[1] && d->Cin >= r.lo[2] && d->Cin <= r.hi[2] &&
        (r.kh < 0 || r.kh == d->KH) && (r.ph < 0 || r.ph == (int)ph))
      return TileCfg{r.v[0], r.v[1], r.v[2], r.v[3], r.v[4], r.v[5]};
  }
  if (hk.tile_c64.set && d->Cout >= 64 && d->Cout < 128 && d->Cin >= 64 && M >= 65536 && !ph)      // A/B hook: the 64->64 class at full resolution
    return TileCfg{hk.tile_c64.v[0], hk.tile_c64.v[1], hk.tile_c64.v[2], hk.tile_c64.v[3], hk.tile_c64.v[4], hk.tile_c64.v[5]};
  const bool narrow = d->Cin <= 8;
  if (d->Wo < 32) return t;                        // (the 4x16 maps: round 4's entries)
  // The 128x512 network (profiles/r05_tile_sweep_hires.txt, batch 8).  Its launches are 2-8x the size of the 32x128 network's, a
  // launch alone covers the chip several times over, and the entries below win in BOTH regimes.
  if (d->Cout >= 256 && M * d->Cout >= 256L * 128 * 128 && M <= 16384)
    return TileCfg{2, 4, 4, 2, 32, 1};             // 128 px x 128 ch as soon as that is >= 256 workgroups (round 4: from 32768 px): 256->256 at 32x128 x4 alone -22 %, saturated -21 %; 4x4 256->512 at 16x64 x8 -24 % / -26 %
  if (M >= 131072 && d->Cout >= 64 && d->Cout < 128 && !ph) {
    if (narrow) { if (d->KH == 3 && M >= 262144) return TileCfg{2, 2, 4, 2, 32, 0}; }                     // VGG16 conv1_1 (3->64): the ring on 128 px x 64 ch (alone -3..-12 %, saturated -13..-17 %)
    else if (M >= 262144) return TileCfg{4, 1, 4, 4, 32, 1};                                             // 64->64 / 32->64 at 128x512: one wave holds all 64 channels of its 64 px - an A fragment feeds four MFMAs (alone -13..-16 %, saturated -18..-24 %; the 64->64 data gradient +-0)
    else if (d->Cin >= 128 && d->KH == 3 && d->stride == 1) return TileCfg{2, 2, 4, 2, 32, 1};           // 3x3 128->64 at 64x256 (decoder): alone -2 %, saturated -14 %
  }
  if (M >= 262144 && narrow && d->Cout > 16 && d->Cout < 64 && d->KH >= 7)
    return TileCfg{8, 1, 4, 2, 32, 0};             // 7x7 3->32 at 128x512, forward and the 32->3 layers' data gradient: the ring on 512 px (alone -7..-12 %, saturated -17..-22 %)
  if (M >= 262144) return t;                       // (the rest of the 128x512 network: round 4's entries)
  if (d->Cout >= 64) {
    if (narrow) return t;
    if (d->Cout >= 128 && d->Cout < 256 && M == 16384) t = TileCfg{1, 8, 4, 1, 32, 1};   // 64->128 / 128->128 at 16x64, batch 16, and their transposes: 64 px x 128 ch (alone -4..-8 %, saturated -26..-30 %)
    else if (d->Cout < 128 && M > 16384) {
      if (ph) t = TileCfg{2, 2, 4, 2, 32, 1};                                            // stride-2 data gradients by phases -> 64 channels (alone +-0, saturated -22..-29 %)
      else if (d->stride != 1 || d->dilate != 1 || d->KH != 3) {}                         // (stride-2 layers and their zero-stuffed gradients: round 4's entries)
      else if (d->Cin <= 32 && M >= 65536) t = TileCfg{8, 1, 4, 2, 32, 0};               // 3x3 32->64 at full resolution (decoder data gradient: alone -10 %, saturated -21 %)
      else if (d->Cin <= 32 && M <= 32768) t = TileCfg{1, 4, 4, 1, 32, 1};               // 3x3 32->64 at 16x64 (alone -3 %, saturated -12 %)
    }
  } else if (d->Cout > 16) {
    if (M >= 65536 && !hk.tile_c32.set) {
      if (narrow && d->want_stats) t = TileCfg{8, 1, 4, 2, 32, 0};                       // 7x7 3->32 forward: the ring on 512 px (alone +-0, saturated -31 %)
      else if (!narrow && !ph && d->KH == 3) t = TileCfg{4, 2, 4, 1, 32, 1};             // 3x3 64->32 at full resolution: direct-B 256 px x 32 ch (alone +1 %, saturated -12 %)
    }
  } else {
    if (M >= 65536 && !ph && d->KH >= 7 && !hk.tile_c16.set) t = TileCfg{8, 1, 4, 1, 32, 1};   // 7x7 32->3: 512 px x 16 ch (alone -8 %, saturated -33 %)
  }
  return t;
}

TileCfg choose_tile_r4(const hdrsky_conv_desc* d) {
  const HdrskyHooks& hk = hdrsky_hooks();
  auto hooked = [](const HdrskyTileHook& h, TileCfg& t) {   // a tuning hook (HDRSKY_EXPERIMENTS=1) replaces the table's entry
    if (h.set) t = TileCfg{h.v[0], h.v[1], h.v[2], h.v[3], h.v[4], h.v[5]};
    return h.set != 0;
  };
  {
    TileCfg o{};
    if (hooked(hk.tile, o)) return o;          // HDRSKY_TILE: every layer on one tile
  }
  // Measured on MI355X (profiles/microbench_conv.py, B=32, 32x128 network): these layers are bound by
  // per-CU operand traffic and instruction issue, not MFMA, so the table prefers 8-wave workgroups
  // (two waves per SIMD) and, for Cout >= 64, the barrier-free direct-B main loop.
  const int tw = (d->Wo >= 32) ? 32 : 16;
  const long M = (long)d->B * d->Ho * d->Wo;
  const bool narrow = d->Cin <= 8;
  TileCfg t{2, 2, 2, 2, tw, 0};
  if (tw == 16) {
    if (d->Cout >= 128) {
      t = TileCfg{1, 8, 2, 1, 16, 1};                                  // 32 px x 128 ch, 8 waves (the 4x16-pixel layers)
      hooked(hk.tile_t16, t);                                          // A/B hook for this class inside the step
    }
    else if (d->Cout >= 64) t = (d->Ho >= 4) ? TileCfg{1, 4, 4, 1, 16, 1} : TileCfg{1, 4, 2, 1, 16, 1};
    else if (d->Cout > 16) t = TileCfg{2, 2, 2, 1, 16, 0};
    else t = TileCfg{4, 1, 1, 1, 16, 0};
  } else if (d->Cout >= 64) {
    // 64 px x 128 ch.  (At up to 4096 pixels - the VGG16 conv3_x layers of a half batch of 16 - that is 128 workgroups and
    // 64 px x 64 ch is faster ALONE (256->256 at 8x32: 16.8 -> 12.5 us, profiles/r03_microbench_tiles_b16.txt), but inside
    // the three-stream step the wider block wins: 2.608 against 2.624 ms, r03_tile_w256_ab.txt - not taken.)
    if (d->Cout >= 256 && M <= 16384) t = TileCfg{1, 8, 4, 1, 32, 1};
    else if (M <= 16384) t = TileCfg{2, 4, 2, 1, 32, 1};               // 64 px x 64 ch, 8 waves
    else t = TileCfg{2, 4, 4, 1, 32, 1};                               // 128 px x 64 ch, 8 waves
    // 128 px x 128 ch - each operand fragment feeds two MFMAs - on >= 32768 pixels x >= 128 channels: 128x512, batch 8: 256->256 at
    // 32x128 74.4 -> 53.8 us, 4x4 256->512 52.3 -> 40.1, the 128x512 training step 5.84 -> 5.72 ms (profiles/r03_tile_wide_ab.txt).
    // (Round 3 kept it out of the table because the distortion-aware data gradient stopped being reproducible beside it; the
    // cause was not this tile but a packed-f32 instruction form in THAT kernel that misbehaves beside any MFMA-dense wave -
    // csrc/Makefile, DESIGN.md section 5.1 - and is gone from the library.)
    if (d->Cout >= 128 && M >= 32768) {
      t = TileCfg{2, 4, 4, 2, 32, 1};
      hooked(hk.tile_wide, t);                                         // A/B hook for this class
    }
  } else if (d->Cout > 16) {
    // 256 px x 32 ch; in BF16X3 the double-buffered hi+lo weight ring of the LDS variant does not fit beside the
    // 7x7 halo planes, so that mode streams the weights per wave as well
    if (M >= 65536) t = (narrow || d->compute == HDRSKY_BF16X3) ? TileCfg{4, 2, 4, 1, 32, 1} : TileCfg{8, 1, 4, 2, 32, 0};
    else t = TileCfg{2, 2, 4, 1, 32, 1};                               // 128 px x 32 ch
    // from 262144 pixels (the 128x512 workload) the direct-B 256 px x 32 ch tile is ahead of the ring variant as well:
    // 7x7 32->32 at 128x512, batch 8: 88 -> 72 us; with the 16-channel class below: 128x512 training step 5.72 -> 5.66 ms,
    // forward + losses 2.99 -> 2.93 ms (profiles/r03_tile_hires_ab.txt)
    if (M >= 262144) t = TileCfg{4, 2, 4, 1, 32, 1};
    if (M >= 65536 && !narrow && d->compute != HDRSKY_BF16X3) hooked(hk.tile_c32, t);   // A/B hook for this class
  } else {
    // Cout <= 16 (the 3-channel output convs).  Round 1 measured the 32-wide column block of the LDS-ring variant (512 px x 32 ch,
    // zero-padded weights) faster than its 16-wide one; the direct-B loop on 256 px x 16 ch beats both: 7x7 32->3 at 32x128,
    // batch 32: 21.6 -> 14.1 us, 3x3 64->3: 15.2 -> 10.7 (profiles/r03_microbench_narrow_out.txt); inside the step -0.5 %, forward
    // pass -0.8 % (r03_tile_c16_ab.txt).  From 262144 pixels (128x512): 512 px x 16 ch (7x7 32->3: 51 -> 39 us).
    if (M >= 262144 && d->compute != HDRSKY_BF16X3) t = TileCfg{8, 1, 4, 1, 32, 1};
    else if (M >= 65536) t = d->compute == HDRSKY_BF16X3 ? TileCfg{4, 2, 4, 1, 32, 1} : TileCfg{4, 1, 4, 1, 32, 1};
    else t = TileCfg{4, 1, 2, 1, 32, 0};
    if (M >= 65536 && d->compute != HDRSKY_BF16X3) hooked(hk.tile_c16, t);
  }
  return t;
}


// ---------------------------------------------------------------------------------------------------
// One output channel (the discriminator's patch-logit conv, discriminator.py:39-40: Conv2D(1, 4) VALID on [B,4,16,512]):
// an implicit GEMM with N = 1 uses one of 16 MFMA columns and one workgroup per sample walks 256 k-steps through the weight
// ring - 32 us for 7 MFLOP.  Here it is what it is, a dot product per output pixel: 16-byte loads of the fp32 input with the
// producer's affine + activation applied on the fly, the filter straight from the packed bf16 image (hi [+ lo]), fp32
// accumulation, wave + block reduction, + bias.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv_dot1_kernel(const ConvKArgs a) {
  // workgroup = one output pixel; thread = items (tap, 8-channel group) tid, tid + 256, ...: every load of a thread is
  // issued before the first use (a first version - workgroup per sample, a wave per output, one item per loop trip - ran
  // at one L2 round trip per item: 61 us)
  __shared__ float sRed[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int ox = bid % a.Wo; bid /= a.Wo;
  const int oy = bid % a.Ho, b = bid / a.Ho;
  const int cin32 = a.Cin >> 5, nq = a.Cin >> 3, nitems = a.ntaps * nq;
  const float* xb = a.x + (size_t)b * a.H * a.W * a.Cin;
  const bool affine = a.in_mode == HDRSKY_IN_AFFINE;
  constexpr int IPT = 4;                                      // items per thread and pass
  float acc = 0.f;
  for (int i0 = tid; i0 < nitems; i0 += 256 * IPT) {
    float4 xa[IPT], xc[IPT], sa[IPT], sc_[IPT], ha[IPT], hc[IPT];
    uint4 wh[IPT], wl[IPT];
    bool ok[IPT];
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      const int i = i0 + u * 256;
      const int ic = min(i, nitems - 1);
      const int tap = ic / nq, qc = ic - tap * nq;
      const int ky = tap / a.KW, kx = tap - ky * a.KW;
      const int iy = oy * a.stride - a.pad_t + ky, ix = ox * a.stride - a.pad_l + kx;
      ok[u] = i < nitems && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const size_t eo = ((size_t)(ok[u] ? iy : 0) * a.W + (ok[u] ? ix : 0)) * a.Cin + qc * 8;
      if (a.x_bf16) {       // a raw conv output stored as bf16: widened here, the affine + activation below as for fp32
        const uint4 b8 = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a.x) + (size_t)b * a.H * a.W * a.Cin + eo);
        xa[u] = make_float4(__builtin_bit_cast(float, b8.x << 16), __builtin_bit_cast(float, b8.x & 0xffff0000u),
                            __builtin_bit_cast(float, b8.y << 16), __builtin_bit_cast(float, b8.y & 0xffff0000u));
        xc[u] = make_float4(__builtin_bit_cast(float, b8.z << 16), __builtin_bit_cast(float, b8.z & 0xffff0000u),
                            __builtin_bit_cast(float, b8.w << 16), __builtin_bit_cast(float, b8.w & 0xffff0000u));
      } else {
        const float* src = xb + eo;
        xa[u] = *reinterpret_cast<const float4*>(src); xc[u] = *reinterpret_cast<const float4*>(src + 4);
      }
      const size_t g = ((size_t)(tap * cin32 + (qc >> 2)) * 4 + (qc & 3)) * a.Npad;   // column n = 0 of the packed image
      wh[u] = a.whi[g];
      if (a.wlo != nullptr) wl[u] = a.wlo[g];
      if (affine) {
        const float* ps = a.in_scale + b * a.ss_bstride + qc * 8;
        const float* ph = a.in_shift + b * a.ss_bstride + qc * 8;
        sa[u] = *reinterpret_cast<const float4*>(ps); sc_[u] = *reinterpret_cast<const float4*>(ps + 4);
        ha[u] = *reinterpret_cast<const float4*>(ph); hc[u] = *reinterpret_cast<const float4*>(ph + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      if (!ok[u]) continue;
      const float in[8] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w, xc[u].x, xc[u].y, xc[u].z, xc[u].w};
      const float sc8[8] = {sa[u].x, sa[u].y, sa[u].z, sa[u].w, sc_[u].x, sc_[u].y, sc_[u].z, sc_[u].w};
      const float sh8[8] = {ha[u].x, ha[u].y, ha[u].z, ha[u].w, hc[u].x, hc[u].y, hc[u].z, hc[u].w};
      const unsigned hw[4] = {wh[u].x, wh[u].y, wh[u].z, wh[u].w};
      const unsigned lw[4] = {wl[u].x, wl[u].y, wl[u].z, wl[u].w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float w = __builtin_bit_cast(float, (j & 1) ? (hw[j >> 1] & 0xffff0000u) : (hw[j >> 1] << 16));
        if (a.wlo != nullptr) w += __builtin_bit_cast(float, (j & 1) ? (lw[j >> 1] & 0xffff0000u) : (lw[j >> 1] << 16));
        float t = affine ? leaky(in[j] * sc8[j] + sh8[j], a.in_slope) : leaky(in[j], a.in_slope);
        // the arithmetic contract of hdrsky_conv2d_fwd: in the single-product mode the staged activation is a bf16 value (what
        // the MFMA path and this layer's data / weight gradients multiply); the two-plane mode multiplies the fp32 activation
        // by hi + lo - the MFMA path's hi*hi + lo*hi + hi*lo plus the lo*lo term it drops (2^-16 relative)
        if (a.wlo == nullptr) t = bf2f(f2bf(t));
        acc += t * w;
      }
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) sRed[wave] = acc;
  __syncthreads();
  if (tid == 0)
    a.y[((size_t)b * a.Ho + oy) * a.Wo + ox] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]) + (a.bias != nullptr ? a.bias[0] : 0.f);
}

// Stride-2 data gradients (descriptor from hdrsky_conv_desc_init_dgrad: dilate == 2) run by output phases on the un-stuffed
// gradient (conv_igemm_kernel, a.phase) unless per-tile statistics are asked for (their tile count follows the stuffed
// form); HDRSKY_NO_PHASE=1 keeps the zero-stuffed operand (A/B and bit-identity tests).
static bool phase_applies(const hdrsky_conv_desc* d) {
  return d->dilate == 2 && d->Cin > 8 && d->upsample == 1 && d->stride == 1 && !d->want_stats && d->compute != HDRSKY_BF16X3 &&
         !hdrsky_hooks().no_phase;     // (the two-plane mode keeps the zero-stuffed form: no phase instantiations of it)
}
// the problem one phase-form launch tiles: four phase grids of ceil(Ho/2) x ceil(Wo/2) pixels per sample
static hdrsky_conv_desc phase_view(const hdrsky_conv_desc* d) {
  hdrsky_conv_desc v = *d;
  v.B = 4 * d->B; v.Ho = cdiv(d->Ho, 2); v.Wo = cdiv(d->Wo, 2);
  return v;
}

static bool dot1_applies(const hdrsky_conv_desc* d, const float* residual) {
  return d->Cout == 1 && d->Cin >= 32 && (d->Cin % 32) == 0 && d->upsample == 1 && d->dilate == 1 && !d->want_stats && !residual &&
         !(d->x_bf16 && d->compute == HDRSKY_BF16X3) && !d->y_bf16 && d->out_slope == 1.f && !d->final_relu && d->res_mode == 0 &&
         d->in_mode != HDRSKY_IN_PARTIALS && !hdrsky_hooks().no_dot1;
}

}  // namespace

static unsigned long long* g_stamps = nullptr;

extern "C" {

/* debug only: per-workgroup s_memtime stamps of subsequent conv launches (8 x u64 per workgroup); null disables */
void hdrsky_debug_conv_stamps(void* buf) { g_stamps = (unsigned long long*)buf; }

const char* hdrsky_version(void) { return "hdrsky 0.1 (gfx950)"; }

int hdrsky_conv_desc_init(hdrsky_conv_desc* d, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                          int same, int upsample) {
  if (!d || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return HDRSKY_EINVAL;
  if (stride != 1 && stride != 2) return HDRSKY_EINVAL;
  if (upsample != 1 && upsample != 2) return HDRSKY_EINVAL;
  *d = hdrsky_conv_desc{};
  d->B = B; d->H = H; d->W = W; d->Cin = Cin; d->Cout = Cout;
  d->KH = KH; d->KW = KW; d->stride = stride; d->upsample = upsample; d->dilate = 1;
  d->Hc = H * upsample; d->Wc = W * upsample;
  if (same) {
    d->Ho = cdiv(d->Hc, stride); d->Wo = cdiv(d->Wc, stride);
    const int th = (d->Ho - 1) * stride + KH - d->Hc, tw = (d->Wo - 1) * stride + KW - d->Wc;
    d->pad_t = (th > 0 ? th : 0) / 2; d->pad_l = (tw > 0 ? tw : 0) / 2;
  } else {
    if (d->Hc < KH || d->Wc < KW) return HDRSKY_EINVAL;
    d->Ho = (d->Hc - KH) / stride + 1; d->Wo = (d->Wc - KW) / stride + 1;
    d->pad_t = d->pad_l = 0;
  }
  d->compute = HDRSKY_BF16;
  d->in_mode = HDRSKY_IN_NONE; d->in_slope = 1.f; d->out_slope = 1.f; d->in_eps = 1e-3f;
  return HDRSKY_OK;
}

int hdrsky_conv_desc_init_dgrad(hdrsky_conv_desc* d, const hdrsky_conv_desc* f) {
  // data gradient of the conv `f` as a stride-1 conv of the (zero-stuffed, for stride 2) output gradient with
  // the flipped/transposed filter: pad' = K-1-pad, output = f's conv-input domain (the 2x-resized image when
  // f.upsample == 2; follow with hdrsky_up2x_bwd).
  if (!d || !f || f->dilate != 1) return HDRSKY_EINVAL;
  *d = hdrsky_conv_desc{};
  d->B = f->B; d->H = f->Ho; d->W = f->Wo; d->Cin = f->Cout; d->Cout = f->Cin;
  d->KH = f->KH; d->KW = f->KW; d->stride = 1; d->upsample = 1; d->dilate = f->stride;
  d->Hc = (f->Ho - 1) * f->stride + 1; d->Wc = (f->Wo - 1) * f->stride + 1;
  d->pad_t = f->KH - 1 - f->pad_t; d->pad_l = f->KW - 1 - f->pad_l;
  d->Ho = f->Hc; d->Wo = f->Wc;
  d->compute = f->compute;
  d->in_mode = HDRSKY_IN_NONE; d->in_slope = 1.f; d->out_slope = 1.f; d->in_eps = 1e-3f;
  return HDRSKY_OK;
}

size_t hdrsky_conv_packed_elems(int KH, int KW, int Cin, int Cout) {
  return (size_t)(conv_ksteps(KH, KW, Cin) + 1) * 4 * roundup(Cout, 64) * 8;  // + one all-zero k-step
}

int hdrsky_conv_pack_weights(const float* w, int KH, int KW, int Cin, int Cout, int transpose_flip, void* packed_hi,
                             void* packed_lo, void* stream) {
  if (!w || !packed_hi) return HDRSKY_EINVAL;
  if (Cin > 8 && (Cin % 32) != 0) return HDRSKY_EUNSUPPORTED;
  const int ks = conv_ksteps(KH, KW, Cin);
  const int Npad = roundup(Cout, 64);
  const size_t total = (size_t)(ks + 1) * 4 * Npad * 8;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, KH, KW, Cin, Cout, Npad,
                     Cin <= 8 ? 1 : 0, transpose_flip, ks, (unsigned short*)packed_hi, (unsigned short*)packed_lo);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_conv_pack_weights_multi(const void* jobs, int njobs, int total_blocks, void* stream) {
  if (!jobs || njobs <= 0 || total_blocks <= 0) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(pack_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)jobs, njobs);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_conv_kernel_name(const hdrsky_conv_desc* d, char* buf, int n) {
  if (!d || !buf || n <= 0) return HDRSKY_EINVAL;
  if (dot1_applies(d, nullptr)) { snprintf(buf, (size_t)n, "conv_dot1_kernel"); return HDRSKY_OK; }
  hdrsky_conv_desc pv = *d;
  if (phase_applies(d)) pv = phase_view(d);   // (an odd filter over several channel groups falls back at launch: not seen here)
  const TileCfg t = choose_tile(&pv, phase_applies(d));
  snprintf(buf, (size_t)n, "conv_igemm_kernel<%d, %d, %d, %d, %d, %s, %s, %s, %s, false>", t.wm, t.wn, t.mi, t.ni, t.tw,
           d->Cin <= 8 ? "true" : "false", d->compute == HDRSKY_BF16X3 ? "true" : "false", t.db ? "true" : "false",
           phase_applies(d) ? "true" : "false");
  return HDRSKY_OK;
}

int hdrsky_conv_stats_nparts(const hdrsky_conv_desc* d) {
  if (!d) return HDRSKY_EINVAL;
  const TileCfg t = choose_tile(d);
  const int bm = t.wm * t.mi * 16;
  return cdiv(d->Ho, bm / t.tw) * cdiv(d->Wo, t.tw);
}

// [host] 1 when hdrsky_conv2d_fwd_emit can write the transformed operand of this layer: single-product mode, >= 32 input channels,
// stride 1 or 2 without resize / dilation, an operand transform to apply (else x itself is the operand)
int hdrsky_conv2d_emit_supported(const hdrsky_conv_desc* d) {
  if (!d || d->compute != HDRSKY_BF16 || d->Cin <= 8 || (d->Cin % 32) != 0) return 0;
  if (d->upsample != 1 || d->dilate != 1 || (d->stride != 1 && d->stride != 2) || d->Cout == 1) return 0;
  if (d->in_mode == HDRSKY_IN_NONE && d->in_slope == 1.f) return 0;
  if (d->Ho * d->stride < d->H || d->Wo * d->stride < d->W) return 0;     // (VALID geometry: the output blocks' input pixels do not cover the image)
  return 1;
}

struct ConvPair { const void* w_hi2; const void* w_lo2; const float* bias2; const float* in_gamma2; const float* in_beta2;
                  const float* residual2; int x_shared; };
static int conv2d_fwd_impl(const hdrsky_conv_desc* d, const float* x, const void* w_hi, const void* w_lo,
                           const float* bias, const float* in_scale, const float* in_shift, const float* in_part,
                           const float* in_gamma, const float* in_beta, const float* residual, float* y,
                           float* stats_part, void* xb_out, void* stream, const ConvPair* pair = nullptr);

int hdrsky_conv2d_fwd(const hdrsky_conv_desc* d, const float* x, const void* w_hi, const void* w_lo,
                      const float* bias, const float* in_scale, const float* in_shift, const float* in_part,
                      const float* in_gamma, const float* in_beta, const float* residual, float* y,
                      float* stats_part, void* stream) {
  return conv2d_fwd_impl(d, x, w_hi, w_lo, bias, in_scale, in_shift, in_part, in_gamma, in_beta, residual, y, stats_part, nullptr, stream);
}

// hdrsky_conv2d_fwd that ALSO writes xb_out [B,H,W,Cin] bf16 = act(norm(x)), the operand as the matrix cores saw it - what the
// layer's weight gradient (hdrsky_wgrad_job with a final bf16 x) reads, without the hdrsky_act_bf16 launch that used to make it
int hdrsky_conv2d_fwd_emit(const hdrsky_conv_desc* d, const float* x, const void* w_hi, const void* w_lo,
                           const float* bias, const float* in_scale, const float* in_shift, const float* in_part,
                           const float* in_gamma, const float* in_beta, const float* residual, float* y,
                           float* stats_part, void* xb_out, void* stream) {
  if (!xb_out) return HDRSKY_EINVAL;
  if (!hdrsky_conv2d_emit_supported(d)) return HDRSKY_EUNSUPPORTED;
  return conv2d_fwd_impl(d, x, w_hi, w_lo, bias, in_scale, in_shift, in_part, in_gamma, in_beta, residual, y, stats_part, xb_out, stream);
}

// Two layers of identical geometry as ONE launch (round 5: the sky / sun decoder pairs of generator.py:110-156 and their data
// gradients - identical shapes issued twice per step): d->B = 2 x the layers' batch; samples [0, B/2) run on (w_hi, w_lo, bias,
// in_gamma, in_beta, residual), samples [B/2, B) on the *2 set; x_shared != 0: x holds B/2 samples that BOTH layers read (no
// operand transform then); in_scale / in_shift / in_part are tables of the whole batch as for any launch.  The tile is the one a
// launch of B/2 samples takes and a sample's arithmetic does not depend on its neighbours: y, the statistics partials and
// everything downstream are bit-identical to the two separate launches (tests/test_pair_gpu.py).
int hdrsky_conv2d_fwd_pair(const hdrsky_conv_desc* d, const float* x, int x_shared, const void* w_hi, const void* w_lo, const float* bias,
                           const void* w_hi2, const void* w_lo2, const float* bias2, const float* in_scale, const float* in_shift,
                           const float* in_part, const float* in_gamma, const float* in_beta, const float* in_gamma2,
                           const float* in_beta2, const float* residual, const float* residual2, float* y, float* stats_part,
                           void* stream) {
  if (!d || (d->B & 1) || !w_hi2) return HDRSKY_EINVAL;
  if ((bias != nullptr) != (bias2 != nullptr) || (residual != nullptr) != (residual2 != nullptr)) return HDRSKY_EINVAL;
  if (d->compute == HDRSKY_BF16X3 && !w_lo2) return HDRSKY_EINVAL;
  if (d->in_mode == HDRSKY_IN_PARTIALS && (!in_gamma2 || !in_beta2)) return HDRSKY_EINVAL;
  if (x_shared && (d->in_mode != HDRSKY_IN_NONE)) return HDRSKY_EUNSUPPORTED;
  const ConvPair pr{w_hi2, w_lo2, bias2, in_gamma2, in_beta2, residual2, x_shared};
  return conv2d_fwd_impl(d, x, w_hi, w_lo, bias, in_scale, in_shift, in_part, in_gamma, in_beta, residual, y, stats_part, nullptr, stream, &pr);
}

static int conv2d_fwd_impl(const hdrsky_conv_desc* d, const float* x, const void* w_hi, const void* w_lo,
                           const float* bias, const float* in_scale, const float* in_shift, const float* in_part,
                           const float* in_gamma, const float* in_beta, const float* residual, float* y,
                           float* stats_part, void* xb_out, void* stream, const ConvPair* pair) {
  if (!d || !x || !w_hi || !y) return HDRSKY_EINVAL;
  const bool narrow = d->Cin <= 8;
  if (!narrow && (d->Cin % 32) != 0) return HDRSKY_EUNSUPPORTED;
  if (narrow && (d->upsample != 1 || d->dilate != 1)) return HDRSKY_EUNSUPPORTED;
  if (d->upsample == 2 && d->dilate == 2) return HDRSKY_EINVAL;
  const bool precise = d->compute == HDRSKY_BF16X3;
  if (precise && !w_lo) return HDRSKY_EINVAL;
  if (d->in_mode == HDRSKY_IN_AFFINE && (!in_scale || !in_shift)) return HDRSKY_EINVAL;
  if (d->in_mode == HDRSKY_IN_PARTIALS && (!in_part || !in_gamma || !in_beta || d->in_nparts <= 0)) return HDRSKY_EINVAL;
  if (d->want_stats && !stats_part) return HDRSKY_EINVAL;
  // a bf16 operand: a final activation, or (round 4) a raw conv output in front of a norm layer stored as bf16 - the
  // operand transform then runs on the widened values; single-product mode, >= 32 channels, no fused resize
  if (d->x_bf16 && (precise || narrow || d->upsample != 1)) return HDRSKY_EUNSUPPORTED;
  if (d->y_bf16 && (precise || (d->Cout & 3))) return HDRSKY_EUNSUPPORTED;
  if (d->res_mode != 0 && (d->res_mode != 1 || !residual || (d->Cout & 3))) return HDRSKY_EINVAL;
  ConvKArgs a{};
  a.x = x; a.whi = (const uint4*)w_hi; a.wlo = (const uint4*)w_lo; a.bias = bias;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_part = in_part; a.in_gamma = in_gamma; a.in_beta = in_beta;
  a.residual = residual; a.y = y; a.stats = stats_part;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.Npad = roundup(d->Cout, 64);
  a.kzero = conv_ksteps(d->KH, d->KW, d->Cin);
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.upsample = d->upsample; a.dilate = d->dilate; a.Hc = d->Hc; a.Wc = d->Wc;
  a.in_mode = d->in_mode; a.ss_bstride = d->ss_bstride; a.in_nparts = d->in_nparts;
  a.in_eps = d->in_eps; a.in_inv_count = 1.f / (float)(d->H * d->W); a.in_slope = d->in_slope;
  a.out_slope = d->out_slope; a.final_relu = d->final_relu; a.want_stats = d->want_stats;
  a.stamps = g_stamps;
  a.x_bf16 = d->x_bf16; a.y_bf16 = d->y_bf16; a.res_mode = d->res_mode; a.mask_slope = d->mask_slope;
  a.xb_out = (unsigned short*)xb_out;
  hipStream_t s = (hipStream_t)stream;
  hdrsky_conv_desc dt = *d;              // what the tile is chosen on: a paired launch takes the tile of ONE of its layers
  if (pair) {
    a.whi2 = (const uint4*)pair->w_hi2; a.wlo2 = (const uint4*)pair->w_lo2; a.bias2 = pair->bias2;
    a.in_gamma2 = pair->in_gamma2; a.in_beta2 = pair->in_beta2; a.residual2 = pair->residual2;
    a.gsplit = d->B / 2; a.x_shared = pair->x_shared;
    dt.B = d->B / 2;
    if (dot1_applies(d, residual) || xb_out || precise || phase_applies(d) || d->upsample == 2) return HDRSKY_EUNSUPPORTED;
    const TileCfg tp = choose_tile(&dt);
    return narrow ? dispatch_tile_pair<true>(a, tp, s) : dispatch_tile_pair<false>(a, tp, s);
  }
  if (dot1_applies(d, residual)) {       // one output channel: a dot product per pixel, not a 16-column MFMA tile
    a.ntaps = d->KH * d->KW;
    if (!precise) a.wlo = nullptr;
    hipLaunchKernelGGL(conv_dot1_kernel, dim3(a.B * a.Ho * a.Wo), dim3(256), 0, s, a);
    HDRSKY_CHECK_LAUNCH();
    return HDRSKY_OK;
  }
  if (phase_applies(d)) {
    ConvKArgs p = a;
    p.phase = 1; p.KHf = d->KH; p.KWf = d->KW; p.KH = (d->KH + 1) / 2; p.KW = (d->KW + 1) / 2;
    p.dilate = 1; p.Hc = d->H; p.Wc = d->W;            // the operand is the gradient itself
    const hdrsky_conv_desc pv = phase_view(&dt);
    const TileCfg tp = choose_tile(&pv, true);
    const int rc = dispatch_tile<false, false, true>(p, tp, s);
    // the phase form declined (odd filter over several channel groups) or has no instantiation / LDS plan for the tile of
    // its phase grid: the zero-stuffed form below computes the same gradient
    if (rc != HDRSKY_EPHASE_FALLBACK && rc != HDRSKY_EUNSUPPORTED) return rc;
  }
  const TileCfg t = choose_tile(&dt);
  if (d->upsample == 2) {
    if (narrow || a.xb_out != nullptr) return HDRSKY_EUNSUPPORTED;
    return precise ? dispatch_tile_up<true>(a, t, s) : dispatch_tile_up<false>(a, t, s);
  }
  if (narrow) return precise ? dispatch_tile<true, true>(a, t, s) : dispatch_tile<true, false>(a, t, s);
  if (a.xb_out != nullptr) return dispatch_tile<false, false, false, true>(a, t, s);      // (hdrsky_conv2d_emit_supported: never narrow / precise)
  return precise ? dispatch_tile<false, true>(a, t, s) : dispatch_tile<false, false>(a, t, s);
}

}  // extern "C"
