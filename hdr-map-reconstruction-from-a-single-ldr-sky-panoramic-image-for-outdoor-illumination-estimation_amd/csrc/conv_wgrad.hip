// Convolution weight gradient on gfx950 matrix cores:
//   dW[ky,kx,ci,co] = sum_{b,oy,ox} X'[b, oy*s+ky-pt, ox*s+kx-pl, ci] * dY[b,oy,ox,co]      db[co] = sum dY
// where X' is the SAME fused operand the forward conv consumed (producer's IN/BN + activation and, for the
// resize-deconv, the 2x bilinear resize, applied while staging).  tf.GradientTape through tf.nn.conv2d
// (train.py:402-406 for ops.py:41-42,121-124 and the Keras Conv2D layers of discriminator.py / sunrad_net.py).
//
// GEMM view: D[ci][co] += A[ci][pixel] * B[pixel][co], reduction over output pixels.  Both MFMA operands need
// 8 consecutive PIXELS of one channel per lane, but the tensors are NHWC (channels contiguous).  The tiles are
// therefore staged pixel-major in LDS ([pixel][channel] bf16 rows) and read with the CDNA4 transposing LDS load
// ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-per-lane): a filter tap is
// then just a ROW offset of the X tile, always 8-byte aligned.
// One workgroup owns a [taps x CB x OB] block of dW, walks `tiles_per_wg` output-pixel tiles accumulating in
// registers (each wave a subset of the taps), and adds its block to dW with fp32 atomics (dW zeroed by the caller).
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

struct WgradArgs {
  const float* x;
  const float* dy;
  float* dw;
  float* db;
  const float* in_scale;
  const float* in_shift;
  const float* in_part;
  const float* in_gamma;
  const float* in_beta;
  int B, H, W, Cin, Ho, Wo, Cout;
  int KH, KW, stride, pad_t, pad_l, upsample, Hc, Wc;
  int in_mode, ss_bstride, in_nparts;
  float in_eps, in_inv_count, in_slope;
  int tiles_x, tiles_y, ntiles, tiles_per_wg, cblocks, oblocks;
  int HT, WT, NPIX, wt_magic, kw_magic, ntaps;
  int RX, RY;                    // LDS row strides (bytes) of the X / dY tiles
  int off_xlo, off_y, off_ylo, off_ss, off_red;
  int wg_target;                 // tuning hook (HDRSKY_WGRAD): workgroups to aim for, 0 = default
};

__device__ __forceinline__ uint2 lds_tr(const unsigned char* p) {
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}

template <int TPW, int CBF, int OBF, int TW, bool NARROW, bool PRECISE, int NW>
__global__ void __launch_bounds__(NW * 64) conv_wgrad_kernel(const WgradArgs a) {
  constexpr int CB = CBF * 16, OB = OBF * 16;
  constexpr int NT = NW * 64;
  constexpr int CBH = (NW == 8) ? CBF / 2 : CBF;   // 8 waves: 4 tap groups x 2 halves of the ci fragments
  static_assert(NW == 4 || (NW == 8 && (CBF % 2) == 0), "8-wave variant needs an even number of ci fragments");
  constexpr int BM = 128, TH = BM / TW;      // output pixels per tile (4 k-steps of 32)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sX = smem;
  unsigned char* sXl = smem + a.off_xlo;
  unsigned char* sY = smem + a.off_y;
  unsigned char* sYl = smem + a.off_ylo;
  float* sScale = reinterpret_cast<float*>(smem + a.off_ss);
  float* sShift = sScale + CB;
  float* sRed = reinterpret_cast<float*>(smem + a.off_red);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, kq = g, lr = lane & 15;
  const int nblk = a.cblocks * a.oblocks;
  const int blk = blockIdx.x % nblk, chunk = blockIdx.x / nblk;
  const int cb0 = (blk / a.oblocks) * CB, ob0 = (blk % a.oblocks) * OB;
  const float slope = a.in_slope;

  const int tgrp = wave & 3;                        // tap group of this wave
  const int cih = (NW == 8) ? (wave >> 2) * CBH : 0;  // first ci fragment of this wave
  f32x4_t acc[TPW][CBH][OBF];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < CBH; ++i)
#pragma unroll
      for (int j = 0; j < OBF; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

  const int tile0 = chunk * a.tiles_per_wg;
  const int tile1 = min(a.ntiles, tile0 + a.tiles_per_wg);
  int prev_b = -1;
  for (int tile = tile0; tile < tile1; ++tile) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * a.stride - a.pad_t, ix0 = ox0 * a.stride - a.pad_l;
    __syncthreads();  // previous tile fully consumed
    if (b != prev_b) {  // per-sample operand transform tables for this block's CB channels
      prev_b = b;
      for (int c = tid; c < CB; c += NT) {
        const int cc = cb0 + c;
        float sc = 1.f, sh = 0.f;
        if (cc < a.Cin) {
          if (a.in_mode == HDRSKY_IN_AFFINE) {
            sc = a.in_scale[b * a.ss_bstride + cc];
            sh = a.in_shift[b * a.ss_bstride + cc];
          } else if (a.in_mode == HDRSKY_IN_PARTIALS) {
            float s = 0.f, ss = 0.f;
            const float* pp = a.in_part + (size_t)b * a.in_nparts * 2 * a.Cin + cc;
            for (int k = 0; k < a.in_nparts; ++k) { s += pp[(2 * k) * a.Cin]; ss += pp[(2 * k + 1) * a.Cin]; }
            const float mean = s * a.in_inv_count;
            const float var = fmaxf(ss * a.in_inv_count - mean * mean, 0.f);
            sc = a.in_gamma[cc] / sqrtf(var + a.in_eps);
            sh = a.in_beta[cc] - mean * sc;
          }
        }
        sScale[c] = sc; sShift[c] = sh;
      }
      __syncthreads();
    }
    // ---- stage X' halo tile: [pixel][CB] bf16 rows (transform + optional 2x bilinear resize) ----------
    {
      constexpr int NQ = CB / 8;
      const int nitems = a.NPIX * NQ;
      for (int i = tid; i < nitems; i += NT) {
        const int px = i / NQ, qc = i % NQ;
        const int hy = (int)(((unsigned)px * (unsigned)a.wt_magic) >> 24);
        const int hx = px - hy * a.WT;
        const int cy = iy0 + hy, cx = ix0 + hx;
        const bool ok = cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
        float v[8];
        if (NARROW) {
          const float* src = a.x + ((size_t)(b * a.H + (ok ? cy : 0)) * a.W + (ok ? cx : 0)) * a.Cin;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = qc * 8 + j;
            const bool cok = ok && c < a.Cin;
            const float t = src[cok ? c : 0];
            v[j] = cok ? leaky(t * sScale[c] + sShift[c], slope) : 0.f;
          }
        } else {
          const int c0 = cb0 + qc * 8;
          const float* xb = a.x + (size_t)b * a.H * a.W * a.Cin + c0;
          if (a.upsample == 2) {
            const float sy = (cy + 0.5f) * 0.5f - 0.5f, sx = (cx + 0.5f) * 0.5f - 0.5f;
            const float fy = floorf(sy), fx = floorf(sx);
            const int ylo = min(max((int)fy, 0), a.H - 1), yhi = max(min((int)ceilf(sy), a.H - 1), 0);
            const int xlo = min(max((int)fx, 0), a.W - 1), xhi = max(min((int)ceilf(sx), a.W - 1), 0);
            const float ly = sy - fy, lx = sx - fx;
            const float* s00 = xb + ((size_t)ylo * a.W + xlo) * a.Cin;
            const float* s01 = xb + ((size_t)ylo * a.W + xhi) * a.Cin;
            const float* s10 = xb + ((size_t)yhi * a.W + xlo) * a.Cin;
            const float* s11 = xb + ((size_t)yhi * a.W + xhi) * a.Cin;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float sc = sScale[qc * 8 + j], sh = sShift[qc * 8 + j];
              const float tl = leaky(s00[j] * sc + sh, slope), tr = leaky(s01[j] * sc + sh, slope);
              const float bl = leaky(s10[j] * sc + sh, slope), br = leaky(s11[j] * sc + sh, slope);
              const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
              v[j] = ok ? top + (bot - top) * ly : 0.f;
            }
          } else {
            const float* src = xb + ((size_t)(ok ? cy : 0) * a.W + (ok ? cx : 0)) * a.Cin;
            const float4 va = *reinterpret_cast<const float4*>(src);
            const float4 vb = *reinterpret_cast<const float4*>(src + 4);
            const float in[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? leaky(in[j] * sScale[qc * 8 + j] + sShift[qc * 8 + j], slope) : 0.f;
          }
        }
        uint4 hi, lo;
        pack8<PRECISE>(v, hi, lo);
        *reinterpret_cast<uint4*>(sX + (size_t)px * a.RX + qc * 16) = hi;
        if (PRECISE) *reinterpret_cast<uint4*>(sXl + (size_t)px * a.RX + qc * 16) = lo;
      }
    }
    // ---- stage dY tile: [pixel][OB] bf16 rows (+ per-thread bias-gradient partial sums) -----------------
    {
      constexpr int NQ = OB / 8;  // NT % NQ == 0: a thread always handles the same 8 output channels
      const int qc = tid % NQ;
      const int n0 = ob0 + qc * 8;
      for (int i = tid; i < BM * NQ; i += NT) {
        const int m = i / NQ;
        const int oy = oy0 + m / TW, ox = ox0 + m % TW;
        const bool ok = oy < a.Ho && ox < a.Wo;
        float v[8];
        const float* src = a.dy + ((size_t)(b * a.Ho + (ok ? oy : 0)) * a.Wo + (ok ? ox : 0)) * a.Cout;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool cok = ok && (n0 + j) < a.Cout;
          v[j] = cok ? src[n0 + j] : 0.f;
          bsum[j] += v[j];
        }
        uint4 hi, lo;
        pack8<PRECISE>(v, hi, lo);
        *reinterpret_cast<uint4*>(sY + (size_t)m * a.RY + qc * 16) = hi;
        if (PRECISE) *reinterpret_cast<uint4*>(sYl + (size_t)m * a.RY + qc * 16) = lo;
      }
    }
    __syncthreads();
    // ---- MFMA: 4 k-steps of 32 output pixels ------------------------------------------------------------
#pragma unroll 1
    for (int r = 0; r < BM / 32; ++r) {
      const int m = r * 32 + g * 8 + q;        // this lane's row of the first 4-pixel block (second: m+4)
      const int mty = m / TW, mtx = m % TW;
      uint4 bh[OBF], bl[OBF];
#pragma unroll
      for (int j = 0; j < OBF; ++j) {
        const unsigned char* ad = sY + (size_t)m * a.RY + (j * 16 + p * 4) * 2;
        const uint2 v0 = lds_tr(ad), v1 = lds_tr(ad + 4 * a.RY);
        bh[j] = uint4{v0.x, v0.y, v1.x, v1.y};
        if (PRECISE) {
          const unsigned char* al = sYl + (size_t)m * a.RY + (j * 16 + p * 4) * 2;
          const uint2 w0 = lds_tr(al), w1 = lds_tr(al + 4 * a.RY);
          bl[j] = uint4{w0.x, w0.y, w1.x, w1.y};
        }
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tap = tgrp + 4 * t;   // wave-uniform
        if (tap < a.ntaps) {
          const int ky = (tap * a.kw_magic) >> 16, kx = tap - ky * a.KW;
          const int xpix = (mty * a.stride + ky) * a.WT + mtx * a.stride + kx;
          const int step4 = 4 * a.stride * a.RX;
#pragma unroll
          for (int i = 0; i < CBH; ++i) {
            const unsigned char* ad = sX + (size_t)xpix * a.RX + ((cih + i) * 16 + p * 4) * 2;
            const uint2 v0 = lds_tr(ad), v1 = lds_tr(ad + step4);
            const uint4 ah = uint4{v0.x, v0.y, v1.x, v1.y};
            uint4 al = uint4{0, 0, 0, 0};
            if (PRECISE) {
              const unsigned char* adl = sXl + (size_t)xpix * a.RX + ((cih + i) * 16 + p * 4) * 2;
              const uint2 w0 = lds_tr(adl), w1 = lds_tr(adl + step4);
              al = uint4{w0.x, w0.y, w1.x, w1.y};
            }
#pragma unroll
            for (int j = 0; j < OBF; ++j) {
              if (PRECISE) {
                acc[t][i][j] = mfma16(al, bh[j], acc[t][i][j]);
                acc[t][i][j] = mfma16(ah, bl[j], acc[t][i][j]);
              }
              acc[t][i][j] = mfma16(ah, bh[j], acc[t][i][j]);
            }
          }
        }
      }
    }
  }

  // ---- epilogue: add this block's partial dW (and db) to global memory --------------------------------------
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = tgrp + 4 * t;
    if (tap < a.ntaps) {
#pragma unroll
      for (int i = 0; i < CBH; ++i)
#pragma unroll
        for (int j = 0; j < OBF; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ci = cb0 + (cih + i) * 16 + kq * 4 + e, co = ob0 + j * 16 + lr;
            if (ci < a.Cin && co < a.Cout) atomicAdd(a.dw + ((size_t)tap * a.Cin + ci) * a.Cout + co, acc[t][i][j][e]);
          }
    }
  }
  if (a.db != nullptr && cb0 == 0) {
    constexpr int NQ = OB / 8;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) sRed[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < OB) {
      const int qc = tid / 8, j = tid % 8;
      float s = 0.f;
      for (int k = qc; k < NT; k += NQ) s += sRed[k * 8 + j];
      if (ob0 + tid < a.Cout) atomicAdd(a.db + ob0 + tid, s);
    }
  }
}

template <int TPW, int CBF, int OBF, int TW, bool NARROW, bool PRECISE>
int launch_wgrad(WgradArgs& a, hipStream_t stream) {
  constexpr int NW = (!NARROW && (CBF % 2) == 0) ? 8 : 4;
  constexpr int CB = CBF * 16, OB = OBF * 16, BM = 128, TH = BM / TW;
  a.tiles_x = cdiv(a.Wo, TW);
  a.tiles_y = cdiv(a.Ho, TH);
  a.ntiles = a.B * a.tiles_x * a.tiles_y;
  a.cblocks = cdiv(NARROW ? 16 : a.Cin, CB);
  a.oblocks = cdiv(a.Cout, OB);
  a.HT = (TH - 1) * a.stride + a.KH;
  a.WT = (TW - 1) * a.stride + a.KW;
  a.NPIX = a.HT * a.WT;
  a.wt_magic = ((1 << 24) + a.WT - 1) / a.WT;
  a.kw_magic = (65536 + a.KW - 1) / a.KW;
  a.ntaps = a.KH * a.KW;
  if (cdiv(a.ntaps, 4) > TPW) return HDRSKY_EUNSUPPORTED;
  if (!NARROW && (a.Cin % CB) != 0) return HDRSKY_EUNSUPPORTED;   // the X staging reads whole CB-channel blocks
  a.RX = CB * 2 + 16;
  a.RY = OB * 2 + 16;
  const int planes = PRECISE ? 2 : 1;
  const int xbytes = roundup(a.NPIX * a.RX, 16), ybytes = roundup(BM * a.RY, 16);
  a.off_xlo = xbytes;
  a.off_y = xbytes * planes;
  a.off_ylo = a.off_y + ybytes;
  a.off_ss = a.off_y + ybytes * planes;
  a.off_red = a.off_ss + 2 * CB * 4;
  const int lds = a.off_red + NW * 64 * 8 * 4;
  if (lds > 160 * 1024) return HDRSKY_EUNSUPPORTED;
  // enough workgroups to fill the chip, few enough that the atomic traffic (one dW block per workgroup) stays small
  const int nblk = a.cblocks * a.oblocks;
  int chunks = cdiv(a.wg_target > 0 ? a.wg_target : 128, nblk);
  if (chunks > a.ntiles) chunks = a.ntiles;
  a.tiles_per_wg = cdiv(a.ntiles, chunks);
  chunks = cdiv(a.ntiles, a.tiles_per_wg);
  auto kern = conv_wgrad_kernel<TPW, CBF, OBF, TW, NARROW, PRECISE, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return HDRSKY_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblk * chunks), dim3(NW * 64), lds, stream, a);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

template <bool NARROW, bool PRECISE>
int dispatch_wgrad(WgradArgs& a, int tpw, int cbf, int obf, int tw, hipStream_t s) {
#define HDRSKY_WG(TPW_, CBF_, OBF_)                                                                   \
  if (tpw == TPW_ && cbf == CBF_ && obf == OBF_) {                                                   \
    return tw == 32 ? launch_wgrad<TPW_, CBF_, OBF_, 32, NARROW, PRECISE>(a, s)                      \
                    : launch_wgrad<TPW_, CBF_, OBF_, 16, NARROW, PRECISE>(a, s);                      \
  }
  if (NARROW) {
    HDRSKY_WG(13, 1, 2) HDRSKY_WG(4, 1, 4) HDRSKY_WG(3, 1, 4)
  } else {
    HDRSKY_WG(3, 2, 4) HDRSKY_WG(3, 4, 4) HDRSKY_WG(3, 4, 2) HDRSKY_WG(3, 2, 2)
    HDRSKY_WG(4, 4, 2) HDRSKY_WG(4, 4, 1) HDRSKY_WG(4, 2, 2)
    HDRSKY_WG(13, 2, 2) HDRSKY_WG(13, 2, 1)
  }
#undef HDRSKY_WG
  return HDRSKY_EUNSUPPORTED;
}

}  // namespace

extern "C" int hdrsky_conv2d_wgrad(const hdrsky_conv_desc* d, const float* x, const float* dy, const float* in_scale,
                                   const float* in_shift, const float* in_part, const float* in_gamma,
                                   const float* in_beta, float* dw, float* db, void* stream) {
  if (!d || !x || !dy || !dw) return HDRSKY_EINVAL;
  if (d->dilate != 1) return HDRSKY_EUNSUPPORTED;
  const bool narrow = d->Cin <= 8;
  if (!narrow && (d->Cin % 32) != 0) return HDRSKY_EUNSUPPORTED;
  if (narrow && d->upsample != 1) return HDRSKY_EUNSUPPORTED;
  if (d->in_mode == HDRSKY_IN_AFFINE && (!in_scale || !in_shift)) return HDRSKY_EINVAL;
  if (d->in_mode == HDRSKY_IN_PARTIALS && (!in_part || !in_gamma || !in_beta || d->in_nparts <= 0)) return HDRSKY_EINVAL;
  WgradArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.db = db;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_part = in_part; a.in_gamma = in_gamma; a.in_beta = in_beta;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.upsample = d->upsample; a.Hc = d->Hc; a.Wc = d->Wc;
  a.in_mode = d->in_mode; a.ss_bstride = d->ss_bstride; a.in_nparts = d->in_nparts;
  a.in_eps = d->in_eps; a.in_inv_count = 1.f / (float)(d->H * d->W); a.in_slope = d->in_slope;
  const int ntaps = d->KH * d->KW;
  const int tpw = ntaps <= 12 ? (ntaps <= 9 ? 3 : 4) : (ntaps <= 16 ? 4 : 13);
  // dW block per workgroup, measured per layer shape (profiles/microbench_wgrad.py): 32x32-channel blocks and
  // ~256 workgroups beat larger blocks - the kernel is bound by staging latency and atomic traffic, not by MFMA
  int cbf, obf;
  if (narrow) { cbf = 1; obf = d->Cout >= 64 ? 4 : (d->Cout >= 32 ? 2 : 1); }
  else { cbf = 2; obf = (tpw == 13 || d->Cout < 32) ? 1 : 2; }
  if (tpw == 13 && obf > 2) obf = 2;               // register budget: TPW*CBF*OBF accumulator fragments per wave
  if (tpw != 13 && !narrow && obf == 1) obf = 2;
  a.wg_target = 256;
  int tw = d->Wo >= 32 ? 32 : 16;
  if (const char* e = getenv("HDRSKY_WGRAD")) {   // tuning hook: "cbf,obf,tw,workgroups"
    int c = 0, o = 0, t = 0, g = 0;
    if (sscanf(e, "%d,%d,%d,%d", &c, &o, &t, &g) >= 3) {
      if (c > 0 && !narrow) cbf = c;
      if (o > 0) obf = o;
      if (t > 0) tw = t;
      a.wg_target = g;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  const bool precise = d->compute == HDRSKY_BF16X3;
  {  // LDS budget: a strided halo tile with 64 channels (x2 planes in BF16X3) can exceed 160 KB -> 32-channel blocks
    const int th = 128 / tw;   // (hook-forced shapes that do not fit return EUNSUPPORTED from the launcher)
    const int npix = ((th - 1) * d->stride + d->KH) * ((tw - 1) * d->stride + d->KW);
    const int planes = precise ? 2 : 1;
    if (!narrow && cbf == 4 && planes * (npix * (64 * 2 + 16) + 128 * (obf * 32 + 16)) > 150 * 1024) cbf = 2;
  }
  if (narrow) return precise ? dispatch_wgrad<true, true>(a, tpw, cbf, obf, tw, s) : dispatch_wgrad<true, false>(a, tpw, cbf, obf, tw, s);
  return precise ? dispatch_wgrad<false, true>(a, tpw, cbf, obf, tw, s) : dispatch_wgrad<false, false>(a, tpw, cbf, obf, tw, s);
}
