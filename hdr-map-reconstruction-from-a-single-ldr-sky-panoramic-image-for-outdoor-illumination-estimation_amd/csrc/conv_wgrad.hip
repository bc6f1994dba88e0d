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

#include <atomic>
#include <cstdarg>
#include <string>

#include "common.h"
#include "hooks.h"

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
  int off_xlo, off_y, off_ylo, off_ss, off_red, off_da;
  int nchunks;                   // pixel split of this job
  int rS;                        // reduce launch in per-job mode (S argument 0): chunk slices per block for this job (4 or 16)
  int x_bf16, dy_bf16;           // operands stored as bf16 (the sample-resident conv chain) instead of fp32
  float* ws;                     // deterministic mode: this job's workspace (per (chunk, block) slabs [taps][CB][OB], then
  float* ws_db;                  //   per (chunk, co block) bias slabs [OB]); null = fp32 atomics straight into dw / db
  // distortion-aware layer (UP == 2): the job is a 1x1 weight gradient over k*k*da_C VIRTUAL input channels - channel
  // (tap, c) of pixel (oy, ox) is the bilinear sample of x[.., c] the layer's gather takes for that tap
  // (distortion_aware_ops.py:62-113), recomputed while staging: the [B,H,W,k*k*C] operand never exists in memory
  const float* da_offs;          // [H][k*k][2] row offsets (hdrsky_da_offsets)
  int da_k, da_C;
};

constexpr int WG_MAXJ = 12;      // jobs per launch (the argument block must stay below 4 KB)
struct MultiArgs {
  int njobs;
  int first[WG_MAXJ + 1];        // first block of each job (prefix sums), first[njobs] = grid size
  int rfirst[WG_MAXJ + 1];       // the same for the reduce launch of the deterministic mode
  WgradArgs job[WG_MAXJ];
};
static_assert(sizeof(MultiArgs) <= 4096, "kernel argument block");

__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {   // 8 bf16 -> 8 floats
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __builtin_bit_cast(float, w[k] << 16);
    v[2 * k + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
  }
}

__device__ __forceinline__ uint2 lds_tr(const unsigned char* p) {
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}

// One launch serves up to WG_MAXJ independent layers ("jobs", passed by value in the kernel argument block): a
// training step has ~40 conv layers whose weight gradients are mutually independent and individually far too small
// for 256 CUs - launched one by one each needs a deep pixel split (fp32 atomic traffic = split x |dW|, ~1 TB/s
// chip-wide) to fill the chip; launched together they fill it with a shallow split and large dW blocks.
// Geometry (template): NW waves = NTG tap groups x CS splits of the ci fragments x OS splits of the co fragments;
// the staging phase (global -> registers -> transform -> bf16 -> LDS) is issue-bound VALU work shared by all
// NW*64 threads.  XR: rows of the X' register prefetch array (items per thread; UP: 4 source pixels per item).
// UP = 1: the forward conv resized its input 2x (resize-deconv): an X' item is interpolated from four source pixels.
// UP = 2: distortion-aware layer: an X' item (pixel, 8 virtual channels of one tap) is the layer's bilinear sample.
template <int TPW, int CBF, int OBF, int TW, bool NARROW, bool PRECISE, int UP, int NW, int CS, int OS, int XR, int BM>
__global__ void __launch_bounds__(NW * 64) conv_wgrad_kernel(const MultiArgs m) {
  constexpr int NT = NW * 64;
  constexpr int CB = CBF * 16, OB = OBF * 16;
  static_assert(CBF % CS == 0 && OBF % OS == 0 && NW % (CS * OS) == 0, "wave roles");
  constexpr int CBH = CBF / CS, OBH = OBF / OS;    // fragments per wave
  constexpr int NTG = NW / (CS * OS);              // tap groups
  constexpr int TH = BM / TW;                      // BM output pixels per tile (BM/32 k-steps): 128, or 64 for 4-row images
  constexpr int NQX = CB / 8, NQY = OB / 8;        // 8-channel groups per staged pixel
  constexpr int XI = UP ? XR / 4 : XR;             // X' items per thread and tile
  int job = 0;
  while (job + 1 < m.njobs && (int)blockIdx.x >= m.first[job + 1]) ++job;
  const WgradArgs& a = m.job[job];
  const int bid = blockIdx.x - m.first[job];
  constexpr int YMAX = (BM * NQY + NT - 1) / NT;   // dY items per thread and tile
  static_assert(NT % NQY == 0, "a thread must always stage the same 8 output channels (bias partial sums)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sX = smem;
  unsigned char* sXl = smem + a.off_xlo;
  unsigned char* sY = smem + a.off_y;
  unsigned char* sYl = smem + a.off_ylo;
  float* sTab = reinterpret_cast<float*>(smem + a.off_ss);   // [sample of this workgroup][scale CB | shift CB]
  float* sRed = reinterpret_cast<float*>(smem + a.off_red);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, kq = g, lr = lane & 15;
  const int nblk = a.cblocks * a.oblocks;
  const int blk = bid % nblk, chunk = bid / nblk;
  const int cb0 = (blk / a.oblocks) * CB, ob0 = (blk % a.oblocks) * OB;
  const float slope = a.in_slope;
  const bool xform = a.in_mode != HDRSKY_IN_NONE || slope != 1.f;

  const int tgrp = wave % NTG;                           // tap group of this wave
  const int cih = ((wave / NTG) % CS) * CBH;             // first ci fragment of this wave
  const int ojh = ((wave / NTG) / CS) * OBH;             // first co fragment of this wave
  f32x4_t acc[TPW][CBH][OBH];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < CBH; ++i)
#pragma unroll
      for (int j = 0; j < OBH; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

  // byte offset of each of this wave's taps inside the X' tile (wave-uniform, tile-independent): computed once instead
  // of per (k-step, tap) in the MFMA phase, which is instruction-issue bound
  int tapoff[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = min(tgrp + NTG * t, a.ntaps - 1);
    const int ky = (tap * a.kw_magic) >> 16, kx = tap - ky * a.KW;
    tapoff[t] = (ky * a.WT + kx) * a.RX;
  }
  const int step4 = 4 * a.stride * a.RX;
  const int tile0 = chunk * a.tiles_per_wg;
  const int tile1 = min(a.ntiles, tile0 + a.tiles_per_wg);
  const int tps = a.tiles_x * a.tiles_y;            // tiles per sample
  const int bfirst = tile0 / tps;
  const int nitems = a.NPIX * NQX;

  // Software pipeline: the global loads of tile t+1 are issued into registers (xr / yr) before the MFMA phase of
  // tile t and are transformed + written to LDS at the top of the next iteration.
  float xr[XR][8];   // X' items of this thread; UP: the four source pixels of its one item
  float yr[YMAX][8];
  float wr[UP == 2 ? XI : 1][4];   // UP == 2: bilinear weights of the four source pixels

  // UP == 2: the block's virtual channels belong to ONE filter tap (CB <= da_C); its sampling offsets of every image row go
  // to LDS once, so that the gather of a tile starts from an LDS read instead of a dependent global load (the three
  // dependent round trips per item - offsets, then corners - were what a tile visit cost: 17 us for 2 items per thread)
  float* sOffDA = reinterpret_cast<float*>(smem + a.off_da);
  if (UP == 2) {
    const int tap = cb0 / a.da_C, k2 = a.da_k * a.da_k;
    for (int i = tid; i < a.H * 2; i += NT) sOffDA[i] = a.da_offs[((i >> 1) * k2 + tap) * 2 + (i & 1)];
    __syncthreads();   // read by the first tile's load phase, which runs before any other barrier
  }
  // ---- per-sample operand transform tables for every sample this workgroup touches ----------------------------
  {
    const int blast = (tile1 - 1) / tps;
    const int ntab = (blast - bfirst + 1) * CB;
    for (int idx = tid; idx < ntab; idx += NT) {
      const int sb = idx / CB, c = idx % CB, b = bfirst + sb, cc = cb0 + c;
      float sc = 1.f, sh = 0.f;
      if (cc < a.Cin) {
        if (a.in_mode == HDRSKY_IN_AFFINE) {
          sc = a.in_scale[b * a.ss_bstride + cc];
          sh = a.in_shift[b * a.ss_bstride + cc];
        } else if (a.in_mode == HDRSKY_IN_PARTIALS) {
          float s0, ss;
          in_partial_sums(a.in_part + (size_t)b * a.in_nparts * 2 * a.Cin + cc, a.in_nparts, a.Cin, s0, ss);
          const float mean = s0 * a.in_inv_count;
          const float var = fmaxf(ss * a.in_inv_count - mean * mean, 0.f);
          sc = a.in_gamma[cc] / sqrtf(var + a.in_eps);
          sh = a.in_beta[cc] - mean * sc;
        }
      }
      sTab[sb * 2 * CB + c] = sc;
      sTab[sb * 2 * CB + CB + c] = sh;
    }
  }

  // (tx, ty, sample) of the tile being staged and of the next one, advanced incrementally: the three runtime
  // divisions per phase and tile they replace are ~30-instruction sequences each on this hardware
  int ctx = 0, cty = 0, cb = 0;                 // tile `tile` (valid once tile >= tile0)
  int ntx = tile0 % a.tiles_x, nty = (tile0 / a.tiles_x) % a.tiles_y, nb = tile0 / tps;   // tile + 1
  for (int tile = tile0 - 1; tile < tile1; ++tile) {
    // ================= stage tile `tile` from the registers filled one iteration ago =============================
    if (tile >= tile0) {
      const int tx = ctx, ty = cty, b = cb;
      const int oy0 = ty * TH, ox0 = tx * TW;
      const int iy0 = oy0 * a.stride - a.pad_t, ix0 = ox0 * a.stride - a.pad_l;
      const float* tsc = sTab + (b - bfirst) * 2 * CB;
      const float* tsh = tsc + CB;
      __syncthreads();  // previous tile fully consumed (first pass: transform tables visible)
      if (UP == 2) {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
          const int i = it * NT + tid;
          if (i < nitems) {
            const int px = i / NQX, qc = i % NQX;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)      // corners outside the image / pixels outside the map carry weight 0
              v[j] = (xr[(it * 4 + 0) % XR][j] * wr[it][0] + xr[(it * 4 + 1) % XR][j] * wr[it][1]) +
                     (xr[(it * 4 + 2) % XR][j] * wr[it][2] + xr[(it * 4 + 3) % XR][j] * wr[it][3]);
            uint4 hi, lo;
            pack8<PRECISE>(v, hi, lo);
            *reinterpret_cast<uint4*>(sX + (size_t)px * a.RX + qc * 16) = hi;
            if (PRECISE) *reinterpret_cast<uint4*>(sXl + (size_t)px * a.RX + qc * 16) = lo;
          }
        }
      } else if (UP == 1) {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
          const int i = it * NT + tid;
          if (i < nitems) {
            const int px = i / NQX, qc = i % NQX;
            const int hy = (int)(((unsigned)px * (unsigned)a.wt_magic) >> 24);
            const int hx = px - hy * a.WT;
            const int cy = iy0 + hy, cx = ix0 + hx;
            const bool ok = cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
            const float sy = (cy + 0.5f) * 0.5f - 0.5f, sx = (cx + 0.5f) * 0.5f - 0.5f;
            const float ly = sy - floorf(sy), lx = sx - floorf(sx);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float sc = tsc[qc * 8 + j], sh = tsh[qc * 8 + j];
              const float tl = leaky(xr[(it * 4 + 0) % XR][j] * sc + sh, slope), tr = leaky(xr[(it * 4 + 1) % XR][j] * sc + sh, slope);
              const float bl = leaky(xr[(it * 4 + 2) % XR][j] * sc + sh, slope), br = leaky(xr[(it * 4 + 3) % XR][j] * sc + sh, slope);
              const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
              v[j] = ok ? top + (bot - top) * ly : 0.f;
            }
            uint4 hi, lo;
            pack8<PRECISE>(v, hi, lo);
            *reinterpret_cast<uint4*>(sX + (size_t)px * a.RX + qc * 16) = hi;
            if (PRECISE) *reinterpret_cast<uint4*>(sXl + (size_t)px * a.RX + qc * 16) = lo;
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
          const int i = it * NT + tid;
          if (i < nitems) {
            const int px = i / NQX, qc = i % NQX;
            const int hy = (int)(((unsigned)px * (unsigned)a.wt_magic) >> 24);
            const int hx = px - hy * a.WT;
            const int cy = iy0 + hy, cx = ix0 + hx;
            const bool ok = cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
            float v[8];
            if (xform) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const bool cok = NARROW ? (ok && qc * 8 + j < a.Cin) : ok;
                v[j] = cok ? leaky(xr[it][j] * tsc[qc * 8 + j] + tsh[qc * 8 + j], slope) : 0.f;
              }
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const bool cok = NARROW ? (ok && qc * 8 + j < a.Cin) : ok;
                v[j] = cok ? xr[it][j] : 0.f;
              }
            }
            uint4 hi, lo;
            pack8<PRECISE>(v, hi, lo);
            *reinterpret_cast<uint4*>(sX + (size_t)px * a.RX + qc * 16) = hi;
            if (PRECISE) *reinterpret_cast<uint4*>(sXl + (size_t)px * a.RX + qc * 16) = lo;
          }
        }
      }
#pragma unroll
      for (int it = 0; it < YMAX; ++it) {   // dY tile: [pixel][OB] bf16 rows (+ per-thread bias-gradient partial sums)
        const int i = it * NT + tid;
        if (i < BM * NQY) {
          const int m = i / NQY, qc = i % NQY;   // NT % NQY == 0: a thread always handles the same 8 output channels
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[j] += yr[it][j];
          uint4 hi, lo;
          pack8<PRECISE>(yr[it], hi, lo);
          *reinterpret_cast<uint4*>(sY + (size_t)m * a.RY + qc * 16) = hi;
          if (PRECISE) *reinterpret_cast<uint4*>(sYl + (size_t)m * a.RY + qc * 16) = lo;
        }
      }
      __syncthreads();
    }
    // ================= issue the global loads of tile + 1 =========================================================
    if (tile + 1 < tile1) {
      const int tx = ntx, ty = nty, b = nb;
      const int oy0 = ty * TH, ox0 = tx * TW;
      const int iy0 = oy0 * a.stride - a.pad_t, ix0 = ox0 * a.stride - a.pad_l;
      if (UP == 2) {
        const int k = a.da_k, pad = (k - 1) / 2, in_h = a.H + k - 1, in_w = a.W + k - 1;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
          const int i = it * NT + tid;
          if (i < nitems) {
            const int px = i / NQX, qc = i % NQX;
            const int hy = (int)(((unsigned)px * (unsigned)a.wt_magic) >> 24);
            const int hx = px - hy * a.WT;
            const int oy = iy0 + hy, ox = ix0 + hx;               // 1x1, stride 1: tile pixel = output pixel
            const bool live = oy < a.H && ox < a.W;
            const int vq = cb0 + qc * 8;                            // first virtual channel of the item
            const int tap = vq / a.da_C, c0 = vq - tap * a.da_C;
            const int tky = tap / k, tkx = tap - tky * k;
            const int oyc = live ? oy : 0;
            const float off_y = sOffDA[oyc * 2], off_x = sOffDA[oyc * 2 + 1];
            const Tap4 sp4 = da_tap((float)(oyc + tky), (float)((live ? ox : 0) + tkx), off_y, off_x, in_h, in_w);
            const int ys[4] = {sp4.y0, sp4.y0, sp4.y1, sp4.y1}, xs[4] = {sp4.x0, sp4.x1, sp4.x0, sp4.x1};
            const float ws4[4] = {sp4.w0, sp4.w1, sp4.w2, sp4.w3};
            const float* xb = a.x + (size_t)b * a.H * a.W * a.da_C + c0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int yy = ys[kk] - pad, xx = xs[kk] - pad;      // back to un-padded coordinates; border = zeros
              const bool in = live && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
              const float* src = xb + (size_t)(in ? yy * a.W + xx : 0) * a.da_C;
              const float4 va = *reinterpret_cast<const float4*>(src);
              const float4 vb = *reinterpret_cast<const float4*>(src + 4);
              float* d = xr[(it * 4 + kk) % XR];
              d[0] = va.x; d[1] = va.y; d[2] = va.z; d[3] = va.w; d[4] = vb.x; d[5] = vb.y; d[6] = vb.z; d[7] = vb.w;
              wr[it][kk] = in ? ws4[kk] : 0.f;
            }
          }
        }
      } else if (UP == 1) {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
          const int i = it * NT + tid;
          if (i < nitems) {
            const int px = i / NQX, qc = i % NQX;
            const int hy = (int)(((unsigned)px * (unsigned)a.wt_magic) >> 24);
            const int hx = px - hy * a.WT;
            const int cy = iy0 + hy, cx = ix0 + hx;
            const float sy = (cy + 0.5f) * 0.5f - 0.5f, sx = (cx + 0.5f) * 0.5f - 0.5f;
            const int ylo = min(max((int)floorf(sy), 0), a.H - 1), yhi = max(min((int)ceilf(sy), a.H - 1), 0);
            const int xlo = min(max((int)floorf(sx), 0), a.W - 1), xhi = max(min((int)ceilf(sx), a.W - 1), 0);
            const float* xb = a.x + (size_t)b * a.H * a.W * a.Cin + cb0 + qc * 8;
            const float* sp[4] = {xb + ((size_t)ylo * a.W + xlo) * a.Cin, xb + ((size_t)ylo * a.W + xhi) * a.Cin,
                                  xb + ((size_t)yhi * a.W + xlo) * a.Cin, xb + ((size_t)yhi * a.W + xhi) * a.Cin};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float4 va = *reinterpret_cast<const float4*>(sp[k]);
              const float4 vb = *reinterpret_cast<const float4*>(sp[k] + 4);
              float* d = xr[(it * 4 + k) % XR];
              d[0] = va.x; d[1] = va.y; d[2] = va.z; d[3] = va.w; d[4] = vb.x; d[5] = vb.y; d[6] = vb.z; d[7] = vb.w;
            }
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < XI; ++it) {
          const int i = it * NT + tid;
          if (i < nitems) {
            const int px = i / NQX, qc = i % NQX;
            const int hy = (int)(((unsigned)px * (unsigned)a.wt_magic) >> 24);
            const int hx = px - hy * a.WT;
            const int cy = iy0 + hy, cx = ix0 + hx;
            const bool ok = cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
            if (NARROW) {
              const float* src = a.x + ((size_t)(b * a.H + (ok ? cy : 0)) * a.W + (ok ? cx : 0)) * a.Cin;
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int c = qc * 8 + j;
                xr[it][j] = src[c < a.Cin ? c : 0];
              }
            } else {
              const size_t eo = ((size_t)(b * a.H + (ok ? cy : 0)) * a.W + (ok ? cx : 0)) * a.Cin + cb0 + qc * 8;
              if (a.x_bf16) {   // job-uniform
                const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a.x) + eo);
                unpack8(u, xr[it]);
              } else {
                const float* src = a.x + eo;
                const float4 va = *reinterpret_cast<const float4*>(src);
                const float4 vb = *reinterpret_cast<const float4*>(src + 4);
                xr[it][0] = va.x; xr[it][1] = va.y; xr[it][2] = va.z; xr[it][3] = va.w;
                xr[it][4] = vb.x; xr[it][5] = vb.y; xr[it][6] = vb.z; xr[it][7] = vb.w;
              }
            }
          }
        }
      }
#pragma unroll
      for (int it = 0; it < YMAX; ++it) {
        const int i = it * NT + tid;
        if (i < BM * NQY) {
          const int m = i / NQY, qc = i % NQY;
          const int oy = oy0 + m / TW, ox = ox0 + m % TW;
          const bool ok = oy < a.Ho && ox < a.Wo;
          const int n0 = ob0 + qc * 8;
          const size_t yo = ((size_t)(b * a.Ho + (ok ? oy : 0)) * a.Wo + (ok ? ox : 0)) * a.Cout;
          const float* src = a.dy + yo;
          if (a.dy_bf16) {    // job-uniform; Cout % 8 == 0 checked on the host
            if (n0 + 8 <= a.Cout) {
              const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a.dy) + yo + n0);
              unpack8(u, yr[it]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) yr[it][j] = (ok && n0 + 8 <= a.Cout) ? yr[it][j] : 0.f;
          } else if ((a.Cout & 7) == 0 && n0 + 8 <= a.Cout) {
            const float4 va = *reinterpret_cast<const float4*>(src + n0);
            const float4 vb = *reinterpret_cast<const float4*>(src + n0 + 4);
            yr[it][0] = ok ? va.x : 0.f; yr[it][1] = ok ? va.y : 0.f; yr[it][2] = ok ? va.z : 0.f; yr[it][3] = ok ? va.w : 0.f;
            yr[it][4] = ok ? vb.x : 0.f; yr[it][5] = ok ? vb.y : 0.f; yr[it][6] = ok ? vb.z : 0.f; yr[it][7] = ok ? vb.w : 0.f;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const bool cok = ok && (n0 + j) < a.Cout;
              const float t = src[cok ? n0 + j : 0];
              yr[it][j] = cok ? t : 0.f;
            }
          }
        }
      }
    }
    ctx = ntx; cty = nty; cb = nb;               // the tile loaded above is staged next
    if (++ntx == a.tiles_x) { ntx = 0; if (++nty == a.tiles_y) { nty = 0; ++nb; } }
    if (tile < tile0) continue;
    // ================= MFMA: 4 k-steps of 32 output pixels =========================================================
#pragma unroll 1
    for (int r = 0; r < BM / 32; ++r) {
      const int m = r * 32 + g * 8 + q;        // this lane's row of the first 4-pixel block (second: m+4)
      const int mty = m / TW, mtx = m % TW;
      const int xbase = ((mty * a.WT + mtx) * a.stride) * a.RX + p * 8;   // + (cih + i) * 32 + tap offset
      uint4 bh[OBH], bl[OBH];
#pragma unroll
      for (int j = 0; j < OBH; ++j) {
        const unsigned char* ad = sY + (size_t)m * a.RY + ((ojh + j) * 16 + p * 4) * 2;
        const uint2 v0 = lds_tr(ad), v1 = lds_tr(ad + 4 * a.RY);
        bh[j] = uint4{v0.x, v0.y, v1.x, v1.y};
        if (PRECISE) {
          const unsigned char* al = sYl + (size_t)m * a.RY + ((ojh + j) * 16 + p * 4) * 2;
          const uint2 w0 = lds_tr(al), w1 = lds_tr(al + 4 * a.RY);
          bl[j] = uint4{w0.x, w0.y, w1.x, w1.y};
        }
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tap = tgrp + NTG * t;   // wave-uniform
        if (tap < a.ntaps) {
#pragma unroll
          for (int i = 0; i < CBH; ++i) {
            const unsigned char* ad = sX + xbase + tapoff[t] + (cih + i) * 32;
            const uint2 v0 = lds_tr(ad), v1 = lds_tr(ad + step4);
            const uint4 ah = uint4{v0.x, v0.y, v1.x, v1.y};
            uint4 al = uint4{0, 0, 0, 0};
            if (PRECISE) {
              const unsigned char* adl = sXl + xbase + tapoff[t] + (cih + i) * 32;
              const uint2 w0 = lds_tr(adl), w1 = lds_tr(adl + step4);
              al = uint4{w0.x, w0.y, w1.x, w1.y};
            }
#pragma unroll
            for (int j = 0; j < OBH; ++j) {
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
    const int tap = tgrp + NTG * t;
    if (tap < a.ntaps) {
#pragma unroll
      for (int i = 0; i < CBH; ++i)
#pragma unroll
        for (int j = 0; j < OBH; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cl = (cih + i) * 16 + kq * 4 + e, ol = (ojh + j) * 16 + lr;
            const int ci = cb0 + cl, co = ob0 + ol;
            if (a.ws != nullptr)   // split-K partial of this (chunk, block): plain stores, summed in a fixed order by wgrad_reduce_kernel
              a.ws[(((size_t)chunk * nblk + blk) * a.ntaps + tap) * (CB * OB) + cl * OB + ol] = acc[t][i][j][e];
            else if (ci < a.Cin && co < a.Cout) atomicAdd(a.dw + ((size_t)tap * a.Cin + ci) * a.Cout + co, acc[t][i][j][e]);
          }
    }
  }
  if (a.db != nullptr && cb0 == 0) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) sRed[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < OB) {
      const int qc = tid / 8, j = tid % 8;
      float s = 0.f;
      for (int k = qc; k < NT; k += NQY) s += sRed[k * 8 + j];
      if (a.ws_db != nullptr) a.ws_db[((size_t)chunk * a.oblocks + blk % a.oblocks) * OB + tid] = s;
      else if (ob0 + tid < a.Cout) atomicAdd(a.db + ob0 + tid, s);
    }
  }
}

// ====================================================================================================================
// v2: weight gradient of layers whose BOTH operands are final bf16 tensors (x' = the activated input the forward conv
// consumed, dY = the gradient w.r.t. its output): the tiles are copied global -> LDS by LDS-DMA (no registers, no VALU
// conversion) into a ring of NS stages, one barrier per tile, the copies of the next NS-1 tiles in flight under the MFMA
// phase of the current one.  The kernel above needs ~6 us per 128-pixel tile for ~1.3 us of MFMA work: global -> registers
// -> convert -> LDS -> barrier -> transposed reads -> barrier, nothing but one register-held tile overlapped.
// Decomposition: workgroup = (dW block of CB x OB channels, tap group of `tgs` filter taps, pixel chunk).  Tap groups
// are whole filter rows where the filter is larger than the group, so a group stages only the halo rows its taps read.
// The 8 waves share the (tap, ci fragment) units of the group round robin; every unit multiplies against all OBF co
// fragments.  LDS rows are padded by one 16-byte slot (the existing kernel's RX = 2 CB + 16 layout and therefore its
// transposed-read addressing); the pad slot is DMA'd from a zero page like every pixel outside the image.
// Partials go to the same [chunk][block][tap][CB][OB] slabs, summed by wgrad_reduce_kernel in a fixed order.
// ====================================================================================================================
__device__ __attribute__((aligned(1024))) unsigned g_zero_page[256];   // zero-initialised: 16 bytes per lane (one line-sized run per piece)

struct Wg2Args {
  const unsigned short* x;       // bf16 [B,H,W,Cin]
  const unsigned short* dy;      // bf16 [B,Ho,Wo,Cout]
  float* ws;                     // partial slabs (as WgradArgs::ws)
  float* ws_db;                  // bias partials, or null
  float* dw;                     // direct != 0 (a job that is not split over pixels): the gradient itself, += (no reduce launch)
  float* db;
  int direct;
  int B, H, W, Cin, Ho, Wo, Cout;
  int KH, KW, stride, pad_t, pad_l, ntaps;
  int tiles_x, tiles_y, ntiles, tiles_per_wg, nchunks, cblocks, oblocks;
  int tgs, ntg;                  // taps per group, groups
  int cbf, obf;                  // 16-channel fragments of the dW block along ci / co (2 or 4 each)
  int TH, tw_shift, BM;          // pixel tile: TH rows x (1 << tw_shift) columns = BM pixels
  int WT, wt_magic;              // halo row length, magic for / WT
  int xbytes, stage_bytes, ns;   // LDS bytes of the X' part of a stage (piece-rounded), of a whole stage, ring depth
};
constexpr int WG2_MAXJ = 12;
struct Multi2Args {
  int njobs;
  int nt;                        // tuning hook HDRSKY_WGRAD2_NT: 1 = the x operand's copies non-temporal, 2 = dy's, 3 = both
  unsigned long long* stamps;    // debug: per-workgroup s_memtime phase stamps (hdrsky_debug_wgrad2_stamps), null in production
  int first[WG2_MAXJ + 1];
  Wg2Args job[WG2_MAXJ];
};
static_assert(sizeof(Multi2Args) <= 4096, "kernel argument block");

__device__ __forceinline__ void wait_vmcnt(int n) {   // n: wave-uniform; larger counts than listed wait for everything (safe)
  switch (n) {
#define HDRSKY_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    HDRSKY_VMC(0) HDRSKY_VMC(1) HDRSKY_VMC(2) HDRSKY_VMC(3) HDRSKY_VMC(4) HDRSKY_VMC(5) HDRSKY_VMC(6) HDRSKY_VMC(7)
    HDRSKY_VMC(8) HDRSKY_VMC(9) HDRSKY_VMC(10) HDRSKY_VMC(11) HDRSKY_VMC(12) HDRSKY_VMC(13) HDRSKY_VMC(14) HDRSKY_VMC(15)
    HDRSKY_VMC(16) HDRSKY_VMC(17) HDRSKY_VMC(18) HDRSKY_VMC(19) HDRSKY_VMC(20) HDRSKY_VMC(21) HDRSKY_VMC(22) HDRSKY_VMC(23)
    HDRSKY_VMC(24) HDRSKY_VMC(25) HDRSKY_VMC(26) HDRSKY_VMC(27) HDRSKY_VMC(28) HDRSKY_VMC(29) HDRSKY_VMC(30) HDRSKY_VMC(31)
    HDRSKY_VMC(32) HDRSKY_VMC(33) HDRSKY_VMC(34) HDRSKY_VMC(35) HDRSKY_VMC(36) HDRSKY_VMC(37) HDRSKY_VMC(38) HDRSKY_VMC(39)
    HDRSKY_VMC(40) HDRSKY_VMC(41) HDRSKY_VMC(42) HDRSKY_VMC(43) HDRSKY_VMC(44) HDRSKY_VMC(45) HDRSKY_VMC(46) HDRSKY_VMC(47)
#undef HDRSKY_VMC
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// One instantiation serves every block shape: cbf / obf (ci / co fragments of the dW block: 2 or 4 each) are job fields, so
// that all eligible layers of a call share ONE launch and one reduce launch.  12 waves: 8 compute waves (two per SIMD) and 4
// loader waves (one per SIMD) that only issue the LDS-DMA copies of the tile NS-1 ahead - the first version had all 8 waves
// issue, then wait, then multiply: per tile 1.4 k cycles of issue + 1.3 k of waiting in front of 3.4 k of MFMA phase
// (profiles/stamp_wgrad2.py).  Everybody meets at one barrier per tile.
template <int UPW>
#ifndef HDRSKY_WG2_MINW
#define HDRSKY_WG2_MINW 3      // waves per SIMD conv_wgrad2_kernel is compiled for (12-wave workgroups: 3 = the whole register file for one workgroup)
#endif
__global__ void __launch_bounds__(768, HDRSKY_WG2_MINW) conv_wgrad2_kernel(const Multi2Args m) {
  constexpr int NWC = 8, NWL = 4, NTC = NWC * 64;
  constexpr int OBFM = 4;                               // co fragments the accumulator array holds (obf <= 4)
  constexpr int MAXPX = 16, MAXPY = 6;                  // DMA pieces per loader wave and tile the slot registers hold
  int job = 0;
  while (job + 1 < m.njobs && (int)blockIdx.x >= m.first[job + 1]) ++job;
  const Wg2Args& a = m.job[job];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, kq = g, lr = lane & 15;
  const int cbf = a.cbf, obf = a.obf;
  const int CB = cbf * 16, OB = obf * 16;
  // 16-byte slots per LDS row: the channels + 2 slots of padding, i.e. row strides of 160 / 96 bytes = 32 (mod 64).  With the
  // k permutation below the 32 lanes that a transposed read serves together touch 8 CONSECUTIVE rows x 32 bytes, and
  // 8 consecutive multiples of such a stride cover the 64 banks exactly once: no bank conflicts
  const int SPX = CB / 8 + 2, SPY = OB / 8 + 2;
  const int RX = SPX * 16, RY = SPY * 16;

  const int nblk = a.cblocks * a.oblocks;
  int l = blockIdx.x - m.first[job];
  const int chunk = l / (nblk * a.ntg);
  l -= chunk * nblk * a.ntg;
  const int tg = l / nblk, blk = l - tg * nblk;
  const int cb0 = (blk / a.oblocks) * CB, ob0 = (blk % a.oblocks) * OB;
  const int t0 = tg * a.tgs, t1 = min(a.ntaps, t0 + a.tgs);
  const int ky0 = t0 / a.KW, ky1 = (t1 - 1) / a.KW;
  const int HTg = (a.TH - 1) * a.stride + (ky1 - ky0) + 1;
  const int NPIXg = HTg * a.WT;
  const int TW = 1 << a.tw_shift;
  const int tile0 = chunk * a.tiles_per_wg;
  const int ntl = min(a.ntiles, tile0 + a.tiles_per_wg) - tile0;
  const int tps = a.tiles_x * a.tiles_y;
  const int ahead = a.ns - 1;
  const bool dbg = m.stamps != nullptr;
  if (dbg && tid == 0) m.stamps[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();

  if (wave >= NWC) {
    // ================================ loader waves ====================================================================
    // DMA pieces (1 KiB = 64 slots) of a stage: X' pieces first, then dY pieces; loader w issues pieces w, w + 4, ...  What
    // a lane copies in piece k is the same in every tile up to the tile's origin: (row, column, channel chunk) of its slot
    // are decoded once into a register per piece (bit 31: a real slot - not padding, not past the tile); per tile it is
    // an add, two bounds tests and the address
    const int lw = wave - NWC;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int px_pieces = (NPIXg * SPX + 63) >> 6, py_pieces = (a.BM * SPY + 63) >> 6;
    const int nx_w = (px_pieces - lw + NWL - 1) / NWL, ny_w = (py_pieces - lw + NWL - 1) / NWL;
    const int n_w = nx_w + ny_w;                        // DMA instructions per tile, this wave
    unsigned xslot[MAXPX], yslot[MAXPY];
#pragma unroll
    for (int k = 0; k < MAXPX; ++k) {
      const int S = (lw + NWL * k) * 64 + lane;
      const int R = S / SPX, c = S - R * SPX;
      const int hy = (int)(((unsigned)R * (unsigned)a.wt_magic) >> 24), hx = R - hy * a.WT;
      xslot[k] = (unsigned)hy | ((unsigned)hx << 10) | ((unsigned)c << 20) | ((c < CB / 8 && R < NPIXg) ? 0x80000000u : 0u);
    }
#pragma unroll
    for (int k = 0; k < MAXPY; ++k) {
      const int S = (lw + NWL * k) * 64 + lane;
      const int mr = S / SPY, c = S - mr * SPY;
      yslot[k] = (unsigned)(mr >> a.tw_shift) | ((unsigned)(mr & (TW - 1)) << 10) | ((unsigned)c << 20) |
                 ((c < OB / 8 && mr < a.BM) ? 0x80000000u : 0u);
    }
    int ib = tile0 / tps;                               // cursor of the next tile to copy, advanced incrementally
    int ity = (tile0 - ib * tps) / a.tiles_x, itx = (tile0 - ib * tps) - ity * a.tiles_x;
    const bool nt_x = (m.nt & 1) != 0, nt_y = (m.nt & 2) != 0;
    auto issue = [&](int st) {
      const int oy0 = ity * a.TH, ox0 = itx * TW;
      const int iy0 = oy0 * a.stride - a.pad_t + ky0, ix0 = ox0 * a.stride - a.pad_l;
      const unsigned sbase = lds0 + (unsigned)st * (unsigned)a.stage_bytes + (unsigned)lw * 1024u;
      const unsigned short* xb = a.x + (size_t)ib * a.H * a.W * a.Cin + cb0;
      const unsigned short* yb = a.dy + (size_t)ib * a.Ho * a.Wo * a.Cout + ob0;
#pragma unroll
      for (int k = 0; k < MAXPX; ++k) {
        if (k < nx_w) {
          const unsigned sl = xslot[k];
          const int cy = iy0 + (int)(sl & 1023u), cx = ix0 + (int)((sl >> 10) & 1023u);
          const bool ok = (int)sl < 0 && (unsigned)cy < (unsigned)a.H && (unsigned)cx < (unsigned)a.W;
          const void* src = ok ? (const void*)(xb + (size_t)(cy * a.W + cx) * a.Cin + ((sl >> 20) & 15u) * 8) : (const void*)(g_zero_page + lane * 4);
          if (nt_x) glds16_nt(src, sbase + (unsigned)k * (NWL * 1024u)); else glds16(src, sbase + (unsigned)k * (NWL * 1024u));
        }
      }
#pragma unroll
      for (int k = 0; k < MAXPY; ++k) {
        if (k < ny_w) {
          const unsigned sl = yslot[k];
          const int oy = oy0 + (int)(sl & 1023u), ox = ox0 + (int)((sl >> 10) & 1023u);
          const bool ok = (int)sl < 0 && oy < a.Ho && ox < a.Wo;
          const void* src = ok ? (const void*)(yb + (size_t)(oy * a.Wo + ox) * a.Cout + ((sl >> 20) & 15u) * 8) : (const void*)(g_zero_page + lane * 4);
          if (nt_y) glds16_nt(src, sbase + (unsigned)a.xbytes + (unsigned)k * (NWL * 1024u)); else glds16(src, sbase + (unsigned)a.xbytes + (unsigned)k * (NWL * 1024u));
        }
      }
      if (++itx == a.tiles_x) { itx = 0; if (++ity == a.tiles_y) { ity = 0; ++ib; } }
    };
    for (int i = 0; i < ahead && i < ntl; ++i) issue(i);
    int st = 0;
    for (int i = 0; i < ntl; ++i) {
      wait_vmcnt(n_w * min(a.ns - 2, ntl - 1 - i));   // this wave's copies of tile i have landed (later tiles may still fly)
      __builtin_amdgcn_s_barrier();                   // tile i is complete; the compute waves are done with tile i - 1's stage
      if (i + ahead < ntl) { int s2 = st + ahead; if (s2 >= a.ns) s2 -= a.ns; issue(s2); }
      if (++st == a.ns) st = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.ws_db != nullptr || (a.direct && a.db != nullptr)) {   // the two barriers of the bias reduction below
      if (cb0 == 0 && tg == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
    }
    return;
  }

  // ================================== compute waves ======================================================================
  // units of this wave: (tap of the group, ci fragment), round robin over the 8 compute waves
  const int nunits = (t1 - t0) * cbf;
  int nmy = 0;
  int tapoff[UPW], utap[UPW], ucif[UPW];
#pragma unroll
  for (int k = 0; k < UPW; ++k) {
    const int u = wave + NWC * k;
    const int uc = min(u, nunits - 1);
    const int tl = uc / cbf, cif = uc - tl * cbf, tap = t0 + tl;
    utap[k] = tap; ucif[k] = cif;
    tapoff[k] = ((tap / a.KW - ky0) * a.WT + tap % a.KW) * RX + cif * 32;
    if (u < nunits) nmy = k + 1;
  }
  f32x4_t acc[UPW][OBFM];
#pragma unroll
  for (int k = 0; k < UPW; ++k)
#pragma unroll
    for (int j = 0; j < OBFM; ++j) acc[k][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const bool do_bias = (a.direct ? a.db != nullptr : a.ws_db != nullptr) && cb0 == 0 && tg == 0;
  float bsum = 0.f;
  // second half of a lane's k values: 16 pixels on - the same tile row when the tile is 32 wide, the next one when 16
  const int step4 = (TW == 32 ? 16 * a.stride : a.WT * a.stride) * RX;
  unsigned long long tw = 0, tc = 0, t_a = 0, t_b = 0, t_d = 0;
  if (dbg && tid == 0) m.stamps[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
  int st = 0;
  for (int i = 0; i < ntl; ++i) {
    if (dbg) t_a = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();                     // tile i has landed (the loaders waited for it)
    if (dbg) t_b = __builtin_amdgcn_s_memtime();
    const unsigned char* sX = smem + (size_t)st * a.stage_bytes;
    const unsigned char* sY = sX + a.xbytes;
    if (do_bias) {
      const int col = tid & (OB - 1);
      for (int mr = tid / OB; mr < a.BM; mr += NTC / OB)
        bsum += bf2f(*reinterpret_cast<const unsigned short*>(sY + (size_t)mr * RY + col * 2));
    }
    // A fragments are read one unit ahead of the MFMAs that consume them (two register sets), the dY fragments of a k-step
    // before its first unit
#pragma unroll 1
    for (int r = 0; r < (a.BM >> 5); ++r) {
      // k permutation: lane group g multiplies pixels {4g..4g+3} and {16+4g..16+4g+3} of the k-step's 32 (both operands
      // alike), so that the two lane groups one transposed read serves together (g = 0,1 | 2,3) read 8 consecutive rows
      const int mm = r * 32 + g * 4 + q;
      const int mty = mm >> a.tw_shift, mtx = mm & (TW - 1);
      const unsigned char* xrow = sX + ((mty * a.WT + mtx) * a.stride) * RX + p * 8;
      uint4 bh[OBFM];
#pragma unroll
      for (int j = 0; j < OBFM; ++j) {
        if (j < obf) {
          const unsigned char* ad = sY + (size_t)mm * RY + (j * 16 + p * 4) * 2;
          const uint2 v0 = lds_tr(ad), v1 = lds_tr(ad + 16 * RY);
          bh[j] = uint4{v0.x, v0.y, v1.x, v1.y};
        }
      }
      uint4 af[2];
      {
        const uint2 v0 = lds_tr(xrow + tapoff[0]), v1 = lds_tr(xrow + tapoff[0] + step4);
        af[0] = uint4{v0.x, v0.y, v1.x, v1.y};
      }
#pragma unroll
      for (int k = 0; k < UPW; ++k) {
        if (k < nmy) {
          if (k + 1 < UPW) {     // (the clamped unit of a wave with fewer units re-reads a valid address; its result is unused)
            const uint2 v0 = lds_tr(xrow + tapoff[k + 1 < UPW ? k + 1 : k]), v1 = lds_tr(xrow + tapoff[k + 1 < UPW ? k + 1 : k] + step4);
            af[(k + 1) & 1] = uint4{v0.x, v0.y, v1.x, v1.y};
          }
#pragma unroll
          for (int j = 0; j < OBFM; ++j)
            if (j < obf) acc[k][j] = mfma16(af[k & 1], bh[j], acc[k][j]);
        }
      }
    }
    if (dbg) { asm volatile("" ::"v"(acc[0][0][0])); t_d = __builtin_amdgcn_s_memtime(); tw += t_b - t_a; tc += t_d - t_b; }
    if (++st == a.ns) st = 0;
  }
  if (dbg && tid == 0) {
    unsigned long long* d = m.stamps + (size_t)blockIdx.x * 8;
    d[2] = __builtin_amdgcn_s_memtime(); d[3] = tw; d[4] = 0; d[5] = tc; d[6] = (unsigned long long)ntl;
  }

  // ---- epilogue: this workgroup's partial of its (chunk, block, taps) slab --------------------------------------------------
#pragma unroll
  for (int k = 0; k < UPW; ++k) {
    if (k < nmy) {
      if (a.direct) {      // the only pixel chunk: this workgroup owns these elements of dW
        float* dst = a.dw + ((size_t)utap[k] * a.Cin + cb0) * a.Cout + ob0;
#pragma unroll
        for (int j = 0; j < OBFM; ++j)
          if (j < obf) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)(ucif[k] * 16 + kq * 4 + e) * a.Cout + j * 16 + lr] += acc[k][j][e];
          }
      } else {
        float* dst = a.ws + (((size_t)chunk * nblk + blk) * a.ntaps + utap[k]) * (CB * OB);
#pragma unroll
        for (int j = 0; j < OBFM; ++j)
          if (j < obf) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(ucif[k] * 16 + kq * 4 + e) * OB + j * 16 + lr] = acc[k][j][e];
          }
      }
    }
  }
  if (do_bias) {
    __builtin_amdgcn_s_barrier();                     // every compute wave is past its last LDS read of the ring
    float* sRed = reinterpret_cast<float*>(smem);
    sRed[tid] = bsum;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < OB) {
      float s2 = 0.f;
      for (int k = tid; k < NTC; k += OB) s2 += sRed[k];
      if (a.direct) a.db[ob0 + tid] += s2;
      else a.ws_db[((size_t)chunk * a.oblocks + blk % a.oblocks) * OB + tid] = s2;
    }
  }
  if (dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) m.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
  }
}


// ====================================================================================================================
// v3: layers with a NARROW side - at most 8 input channels (the 7x7 3->32 stems, the 4x4 stride-2 6->64 first layers of the
// discriminator / sun-radiance stacks) or at most 4 output channels (the 7x7 32->3 decoder tails, the 4x4 512->1
// discriminator head).  On the kernels above such a layer pays a whole 16-row MFMA fragment per filter tap for its 3 useful
// rows (49 fragments for 7x7) and a partial slab of 16 padded channels per tap and workgroup: 75-80 us per layer for 1.2
// GFLOP.  Here the narrow tensor sits in LDS as [row][column][CP channels] bf16 (CP = 4 or 8: 8 / 16 bytes per pixel),
// so that the 32 bytes a transposed read takes per k pixel are 4 (2) NEIGHBOURING pixels x CP channels: an M fragment's
// 16 rows are (kx .. kx+3, channel) of one filter row - 14 fragments for 7x7 x 3, 8 for 4x4 x 6.  Consecutive k pixels read
// overlapping windows (row stride = one pixel), which LDS does not mind.
// One formulation serves both sides.  "wide" = the tensor the k pixels run over, "narrow" = the one read through the tap
// window: narrow pixel = k pixel * stride + tap' - P.
//   Cin <= 8:  wide = dY, narrow = X, tap' = tap, P = SAME padding            D[(tap', ci)][co]
//   Cout <= 4: wide = X' (k = INPUT pixels), narrow = dY, tap' = K-1-tap, P = K-1-pad (stride 1)   D[(tap', co)][ci]
// Both operands go through registers (fp32 -> bf16; the wide one with its producer's normalisation + activation when it
// is the raw fp32 tensor of an X'), prefetched one tile ahead into a double-buffered LDS stage: one barrier per tile.
// Workgroup = (block of <= 64 wide channels, pixel chunk); partial slabs in the reduce launch's [chunk][block][tap][CB][OB]
// layout with (CB, OB) = (CP, wide block) or (wide block, 4): compact, 25 KB per workgroup for 7x7 3->32.
// ====================================================================================================================
struct Wg3Args {
  const void* wide;              // [B,Hk,Wk,Cw] fp32 or bf16
  const float* narrow;           // [B,Hn,Wn,Cn] fp32
  const float* in_scale;         // operand transform of the wide tensor (Cout <= 4 side, fp32 X only)
  const float* in_shift;
  const float* in_part;
  const float* in_gamma;
  const float* in_beta;
  float* ws;
  float* ws_db;
  int B, Hk, Wk, Cw, Hn, Wn, Cn;
  int KH, KW, stride, PT, PL;
  int in_mode, ss_bstride, in_nparts;
  float in_eps, in_inv_count, in_slope;
  int wide_bf16, transposed, bias_mode;   // bias_mode: 0 none, 1 column sums of the wide tile (Cin <= 8), 2 sums of the narrow tensor
  int CP, XPF, NFX, NMF;         // padded narrow channels (4 / 8), columns per fragment (16 / CP), fragments per filter row, KH * NFX
  int NB, NF, nblocks;           // wide channels per workgroup, NB / 16, Cw / NB
  int nsplit, nunits;            // waves per M fragment (each a share of the NF wide fragments), NMF * nsplit
  int TH, tw_shift, BM, tiles_x, tiles_y, ntiles, tiles_per_wg, nchunks;
  int HT, WTn, wt_magic, npixn;  // narrow tile: rows, pixels per row, magic of / WTn, pixels
  int RY, nbytes, stage_bytes, off_ss, nsamp;
  int slabCB, slabOB;
  int deep;                      // host only: one wide item and one narrow pixel per thread and tile -> the D = 4 instantiations
};
constexpr int WG3_MAXJ = 8;
struct Multi3Args {
  int njobs;
  unsigned long long* stamps;    // debug: per-workgroup s_memtime phase stamps (hdrsky_debug_wgrad2_stamps), null in production
  int first[WG3_MAXJ + 1];
  Wg3Args job[WG3_MAXJ];
};
static_assert(sizeof(Multi3Args) <= 4096, "kernel argument block");

// D: tiles whose global loads are all issued before the first of them is consumed (a workgroup walks its chunk in groups of D;
// the first version prefetched one tile ahead and paid a global round trip per 128-pixel tile: 22 us for 4 tiles), WI / NI:
// 8-channel items of the wide tile / pixels of the narrow tile per thread and tile, CPR: registers per narrow pixel (>= the
// job's CP), UPW: (M fragment, wide share) units per wave.  The register arrays hold RAW data (wide: 16 / 32 bytes per item).
template <int D, int WI, int NI, int CPR, int UPW>
__global__ void __launch_bounds__(512, 2) conv_wgrad3_kernel(const Multi3Args m) {
  constexpr int NT = 512, NW = 8;
  static_assert((D & 1) == 0, "the LDS stage of a tile is its parity within the group");
  int job = 0;
  while (job + 1 < m.njobs && (int)blockIdx.x >= m.first[job + 1]) ++job;
  const Wg3Args& a = m.job[job];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, kq = g, lr = lane & 15;
  const int l = blockIdx.x - m.first[job];
  const int chunk = l / a.nblocks, blk = l - chunk * a.nblocks;
  const int n0 = blk * a.NB;
  const int TW = 1 << a.tw_shift;
  const int PB = a.CP * 2;                              // bytes per narrow pixel
  const int tile0 = chunk * a.tiles_per_wg;
  const int ntl = min(a.ntiles, tile0 + a.tiles_per_wg) - tile0;
  const int tps = a.tiles_x * a.tiles_y;
  const int bfirst = tile0 / tps;
  const int NQ = a.NB >> 3;                             // 8-channel items per wide row
  const int nwitems = a.BM * NQ;
  const float slope = a.in_slope;
  const bool xform = a.in_mode != HDRSKY_IN_NONE || slope != 1.f;
  float* sTab = reinterpret_cast<float*>(smem + a.off_ss);
  const bool dbg = m.stamps != nullptr;
  unsigned long long t_ld = 0, t_st = 0, t_bar = 0, t_cp = 0, t_a = 0, t_b = 0;
  if (dbg && tid == 0) m.stamps[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();

  // ---- per-sample transform tables of the wide operand -------------------------------------------------------------------
  if (xform) {
    const int blast = (tile0 + ntl - 1) / tps;
    const int ntab = (blast - bfirst + 1) * a.NB;
    for (int idx = tid; idx < ntab; idx += NT) {
      const int sb = idx / a.NB, c = idx - sb * a.NB, b = bfirst + sb, cc = n0 + c;
      float sc = 1.f, sh = 0.f;
      if (a.in_mode == HDRSKY_IN_AFFINE) {
        sc = a.in_scale[b * a.ss_bstride + cc];
        sh = a.in_shift[b * a.ss_bstride + cc];
      } else if (a.in_mode == HDRSKY_IN_PARTIALS) {
        float s0, ss;
        in_partial_sums(a.in_part + (size_t)b * a.in_nparts * 2 * a.Cw + cc, a.in_nparts, a.Cw, s0, ss);
        const float mean = s0 * a.in_inv_count;
        const float var = fmaxf(ss * a.in_inv_count - mean * mean, 0.f);
        sc = a.in_gamma[cc] / sqrtf(var + a.in_eps);
        sh = a.in_beta[cc] - mean * sc;
      }
      sTab[sb * 2 * a.NB + c] = sc;
      sTab[sb * 2 * a.NB + a.NB + c] = sh;
    }
    __syncthreads();                                    // read by store_tile of the first tile, ahead of any other barrier
  }

  // ---- units of this wave ---------------------------------------------------------------------------------------------------
  const int jcnt = a.NF / a.nsplit;                     // wide fragments per unit (<= 4)
  int nmy = 0;
  const int uj0w = (wave % a.nsplit) * jcnt;             // nsplit divides the wave count: every unit of a wave has the same share
  int tapoff[UPW], umf[UPW];
#pragma unroll
  for (int k = 0; k < UPW; ++k) {
    const int u = wave + NW * k;
    const int uc = min(u, a.nunits - 1);
    const int mf = uc / a.nsplit;
    const int ky = mf / a.NFX, f = mf - ky * a.NFX;
    umf[k] = mf;
    tapoff[k] = ky * a.WTn * PB + f * 32;
    if (u < a.nunits) nmy = k + 1;
  }
  f32x4_t acc[UPW][4];
#pragma unroll
  for (int k = 0; k < UPW; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[k][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

  uint4 wq[D][WI][2];                                   // raw wide items: 8 fp32 (both) or 8 bf16 ([0])
  float nr[D][NI][CPR];
  int torg[D], tsb[D];                                  // (row << 16 | column) of the tile's origin in k pixels, its sample - bfirst
  int ib = bfirst;                                      // cursor of the next tile to load
  int ity = (tile0 - ib * tps) / a.tiles_x, itx = (tile0 - ib * tps) - ity * a.tiles_x;
  int nloaded = 0;

  // what a thread loads / stores is the same in every tile up to the tile's origin: decoded once.  Wide item: (row, column) in
  // the tile, channel offset, LDS byte offset (negative: no item); narrow pixel: (row, column) of the halo tile, LDS offset,
  // bit 30 of the packed position: one of the tile's own pixels (bias sums of a narrow dY)
  int w_pos[WI], w_lds[WI], w_c8[WI], n_pos[NI], n_lds[NI];
#pragma unroll
  for (int it = 0; it < WI; ++it) {
    const int i = it * NT + tid, ic = min(i, nwitems - 1);
    const int mr = ic / NQ, qc = ic - mr * NQ;
    w_pos[it] = ((mr >> a.tw_shift) << 16) | (mr & (TW - 1));
    w_c8[it] = qc * 8;
    w_lds[it] = i < nwitems ? mr * a.RY + qc * 16 : -1;
  }
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int i = it * NT + tid, ic = min(i, a.npixn - 1);
    const int hy = (int)(((unsigned)ic * (unsigned)a.wt_magic) >> 24), hx = ic - hy * a.WTn;
    const bool own = hy >= a.PT && hy < a.PT + a.TH && hx >= a.PL && hx < a.PL + TW;
    n_pos[it] = (hy << 16) | hx | (own ? (1 << 30) : 0);
    n_lds[it] = i < a.npixn ? ic * PB : -1;
  }

  // all loads of a tile, unconditionally (past the chunk's end the last tile is loaded again: straight-line code, so that the
  // compiler's counted waits let the first tile of a group be consumed while the loads of the others are still in flight)
  auto load_tile = [&](int t) {
    const int ky0 = ity * a.TH, kx0 = itx * TW;
    const int ny0 = ky0 * a.stride - a.PT, nx0 = kx0 * a.stride - a.PL;
    torg[t] = (ky0 << 16) | kx0; tsb[t] = ib - bfirst;
#pragma unroll
    for (int it = 0; it < WI; ++it) {
      const int oy = ky0 + (w_pos[it] >> 16), ox = kx0 + (w_pos[it] & 0xffff);
      const bool ok = oy < a.Hk && ox < a.Wk;
      const size_t eo = ((size_t)(ib * a.Hk + (ok ? oy : 0)) * a.Wk + (ok ? ox : 0)) * a.Cw + n0 + w_c8[it];
      if (a.wide_bf16) {
        wq[t][it][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a.wide) + eo);
      } else {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(a.wide) + eo);
        wq[t][it][0] = src[0]; wq[t][it][1] = src[1];
      }
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int cy = ny0 + ((n_pos[it] >> 16) & 0x3fff), cx = nx0 + (n_pos[it] & 0xffff);
      const bool ok = (unsigned)cy < (unsigned)a.Hn && (unsigned)cx < (unsigned)a.Wn;
      const float* src = a.narrow + ((size_t)(ib * a.Hn + (ok ? cy : 0)) * a.Wn + (ok ? cx : 0)) * a.Cn;
#pragma unroll
      for (int j = 0; j < CPR; ++j) {
        const float v = src[j < a.Cn ? j : 0];
        nr[t][it][j] = (ok && j < a.Cn) ? v : 0.f;
      }
    }
    if (++nloaded < ntl) { if (++itx == a.tiles_x) { itx = 0; if (++ity == a.tiles_y) { ity = 0; ++ib; } } }
  };
  auto store_tile = [&](int t, int st) {
    unsigned char* sN = smem + (size_t)st * a.stage_bytes;
    unsigned char* sY = sN + a.nbytes;
    const float* tsc = sTab + tsb[t] * 2 * a.NB;
    const float* tsh = tsc + a.NB;
    const int ky0 = torg[t] >> 16, kx0 = torg[t] & 0xffff;
#pragma unroll
    for (int it = 0; it < WI; ++it) {
      if (w_lds[it] >= 0) {
        const int oy = ky0 + (w_pos[it] >> 16), ox = kx0 + (w_pos[it] & 0xffff);
        const bool ok = oy < a.Hk && ox < a.Wk;
        if (a.wide_bf16 && !xform && a.bias_mode != 1) {     // a final bf16 tensor: copied
          *reinterpret_cast<uint4*>(sY + w_lds[it]) = ok ? wq[t][it][0] : uint4{0, 0, 0, 0};
        } else {
          float v[8];
          if (a.wide_bf16) unpack8(wq[t][it][0], v);
          else {
            const uint4 u0 = wq[t][it][0], u1 = wq[t][it][1];
            v[0] = __builtin_bit_cast(float, u0.x); v[1] = __builtin_bit_cast(float, u0.y); v[2] = __builtin_bit_cast(float, u0.z);
            v[3] = __builtin_bit_cast(float, u0.w); v[4] = __builtin_bit_cast(float, u1.x); v[5] = __builtin_bit_cast(float, u1.y);
            v[6] = __builtin_bit_cast(float, u1.z); v[7] = __builtin_bit_cast(float, u1.w);
          }
          if (xform) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? leaky(v[j] * tsc[w_c8[it] + j] + tsh[w_c8[it] + j], slope) : 0.f;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : 0.f;
          }
          if (a.bias_mode == 1) {          // NT % NQ == 0: a thread always holds the same 8 wide channels
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[j] += v[j];
          }
          uint4 hi, lo;
          pack8<false>(v, hi, lo);
          *reinterpret_cast<uint4*>(sY + w_lds[it]) = hi;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (n_lds[it] >= 0) {
        if (a.bias_mode == 2 && blk == 0 && (n_pos[it] >> 30) != 0) {   // the tile's own pixels (tap' = P), not its halo
#pragma unroll
          for (int j = 0; j < 4; ++j) bsum[j] += nr[t][it][j];
        }
        const unsigned w0 = f2bf(nr[t][it][0]) | ((unsigned)f2bf(nr[t][it][1]) << 16);
        const unsigned w1 = f2bf(nr[t][it][2]) | ((unsigned)f2bf(nr[t][it][3]) << 16);
        if (CPR == 4 || a.CP == 4) *reinterpret_cast<uint2*>(sN + n_lds[it]) = uint2{w0, w1};
        else {
          const unsigned w2 = f2bf(nr[t][it][4 % CPR]) | ((unsigned)f2bf(nr[t][it][5 % CPR]) << 16);
          const unsigned w3 = f2bf(nr[t][it][6 % CPR]) | ((unsigned)f2bf(nr[t][it][7 % CPR]) << 16);
          *reinterpret_cast<uint4*>(sN + n_lds[it]) = uint4{w0, w1, w2, w3};
        }
      }
    }
  };
  // k permutation of conv_wgrad2_kernel: lane group g multiplies pixels {4g..4g+3} and {16+4g..16+4g+3} of a k-step.  With 32-pixel
  // tile rows a k-step is one row (second half 16 columns on), with 16-pixel rows two (second half = the next row): the lane's
  // byte offsets of k-step 0, and what a k-step adds
  const int mm0 = g * 4 + q;
  const int ab0_0 = (mm0 * a.stride) * PB + p * 8;
  const int ab1_0 = ab0_0 + (TW == 32 ? 16 * a.stride : a.WTn * a.stride) * PB;
  const int astep = (TW == 32 ? 1 : 2) * a.WTn * a.stride * PB;
  const int yb_0 = mm0 * a.RY + uj0w * 32 + p * 8;
  auto kstep = [&](const unsigned char* sN, const unsigned char* sY, int r) {
    const unsigned char* yrow = sY + yb_0 + r * 32 * a.RY;
    uint4 bh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < jcnt) {
        const uint2 v0 = lds_tr(yrow + j * 32), v1 = lds_tr(yrow + j * 32 + 16 * a.RY);
        bh[j] = uint4{v0.x, v0.y, v1.x, v1.y};
      }
    }
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
      if (k < nmy) {
        const uint2 a0 = lds_tr(sN + ab0_0 + r * astep + tapoff[k]), a1 = lds_tr(sN + ab1_0 + r * astep + tapoff[k]);
        const uint4 af = uint4{a0.x, a0.y, a1.x, a1.y};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < jcnt) acc[k][j] = mfma16(af, bh[j], acc[k][j]);
      }
    }
  };
  auto compute_tile = [&](int st) {
    const unsigned char* sN = smem + (size_t)st * a.stage_bytes;
    const unsigned char* sY = sN + a.nbytes;
    kstep(sN, sY, 0); kstep(sN, sY, 1);
    if (a.BM == 128) { kstep(sN, sY, 2); kstep(sN, sY, 3); }
  };

  if (dbg && tid == 0) m.stamps[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
  for (int grp = 0; grp < ntl; grp += D) {
    if (dbg) t_a = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int t = 0; t < D; ++t) load_tile(t);
    if (dbg) { t_b = __builtin_amdgcn_s_memtime(); t_ld += t_b - t_a; }
#pragma unroll
    for (int t = 0; t < D; ++t) {
      if (grp + t < ntl) {
        // stage t & 1 was last read by the tile two back; every wave has passed the barrier of the tile in between since
        if (dbg) t_a = __builtin_amdgcn_s_memtime();
        store_tile(t, t & 1);
        if (dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_b = __builtin_amdgcn_s_memtime(); t_st += t_b - t_a; }
        __syncthreads();
        if (dbg) { t_a = __builtin_amdgcn_s_memtime(); t_bar += t_a - t_b; }
        compute_tile(t & 1);
        if (dbg) { asm volatile("" ::"v"(acc[0][0][0])); t_b = __builtin_amdgcn_s_memtime(); t_cp += t_b - t_a; }
      }
    }
  }
  if (dbg && tid == 0) {
    unsigned long long* d = m.stamps + (size_t)blockIdx.x * 8;
    d[2] = __builtin_amdgcn_s_memtime(); d[3] = t_ld; d[4] = t_st; d[5] = t_bar; d[6] = t_cp;
  }

  // ---- epilogue: this workgroup's slab ---------------------------------------------------------------------------------------
  const int ntaps = a.KH * a.KW;
  float* slab = a.ws + ((size_t)chunk * a.nblocks + blk) * ntaps * a.slabCB * a.slabOB;
#pragma unroll
  for (int k = 0; k < UPW; ++k) {
    if (k < nmy) {
      const int ky = umf[k] / a.NFX, f = umf[k] - ky * a.NFX;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int mrow = kq * 4 + e;
        const int xo = f * a.XPF + mrow / a.CP, c = mrow % a.CP;
        if (xo < a.KW) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < jcnt) {
              const int n = (uj0w + j) * 16 + lr;
              if (a.transposed) {
                const int tap = (a.KH - 1 - ky) * a.KW + (a.KW - 1 - xo);
                if (c < a.slabOB) slab[((size_t)tap * a.slabCB + n) * a.slabOB + c] = acc[k][j][e];
              } else {
                slab[((size_t)(ky * a.KW + xo) * a.slabCB + c) * a.slabOB + n] = acc[k][j][e];
              }
            }
          }
        }
      }
    }
  }
  if (a.ws_db != nullptr && (a.bias_mode == 1 || (a.bias_mode == 2 && blk == 0))) {
    // within a wave by shuffles (lanes of equal lane % NQ hold the same 8 channels), across the 8 waves through LDS, in a
    // fixed order (one thread per channel walking all 512 partials was 13-17 k cycles of dependent LDS reads)
    __syncthreads();                                    // every wave is past its last read of the stages
    float* sRed = reinterpret_cast<float*>(smem);
    if (a.bias_mode == 1) {
      for (int o = NQ; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += __shfl_xor(bsum[j], o);
      }
      if (lane < NQ) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sRed[wave * a.NB + lane * 8 + j] = bsum[j];
      }
      __syncthreads();
      if (tid < a.NB) {
        float s2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s2 += sRed[w * a.NB + tid];
        a.ws_db[((size_t)chunk * a.nblocks + blk) * a.NB + tid] = s2;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) bsum[j] = wave_sum(bsum[j]);
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sRed[wave * 4 + j] = bsum[j];
      }
      __syncthreads();
      if (tid < 4) {
        float s2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s2 += sRed[w * 4 + tid];
        a.ws_db[(size_t)chunk * 4 + tid] = s2;
      }
    }
  }
  if (dbg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) m.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
  }
}


// Second stage of the deterministic split-K: dw[tap][ci][co] += sum over the job's pixel chunks of the workgroup
// partials, db likewise - always in the same order.  One launch for the jobs of a conv_wgrad_kernel launch; CB / OB = that
// launch's block size in channels.  A block owns 256/S float4 of a chunk's slab set and S chunk slices: slice q sums the
// chunks q, q+S, ... (16-byte loads, independent chains in flight), then slice 0 adds the S partial sums in slice order.
// S = 4 for shallow splits, 16 when a job is split over >= 32 chunks (the 7x7 / narrow layers: few dW elements, many
// chunks - one chain per element was latency bound there: ~200 blocks of 128 dependent strided loads).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const MultiArgs m, int CB, int OB, int S) {
  __shared__ float4 sPart[15 * 64];
  int job = 0;
  while (job + 1 < m.njobs && (int)blockIdx.x >= m.rfirst[job + 1]) ++job;
  const WgradArgs& a = m.job[job];
  if (CB == 0) { CB = a.RX; OB = a.RY; }               // launch of conv_wgrad2_kernel: the block shape is a job field
  if (S == 0) S = a.rS;                                 // ... and so is the slice count (jobs of several compute launches share this one)
  const int nblk = a.cblocks * a.oblocks, ob4 = OB >> 2;
  const size_t slab = (size_t)a.ntaps * CB * OB;                 // floats of one (chunk, block) partial
  const size_t nvec = (size_t)nblk * a.ntaps * CB * ob4;         // float4 of one chunk
  const int per = 256 / S;                                        // float4 per block
  const int lane = threadIdx.x % per, slice = threadIdx.x / per;
  const int rb = blockIdx.x - m.rfirst[job];
  const size_t v = (size_t)rb * per + lane;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (v < nvec) {
    const float4* src = reinterpret_cast<const float4*>(a.ws) + v;   // slabs of a chunk are contiguous: [blk][tap][cl][ol]
    const size_t cstride = (size_t)nblk * slab / 4;
    int c = slice;
    for (; c + 3 * S < a.nchunks; c += 4 * S) {
      const float4 q0 = src[(size_t)c * cstride], q1 = src[(size_t)(c + S) * cstride];
      const float4 q2 = src[(size_t)(c + 2 * S) * cstride], q3 = src[(size_t)(c + 3 * S) * cstride];
      acc.x += (q0.x + q1.x) + (q2.x + q3.x); acc.y += (q0.y + q1.y) + (q2.y + q3.y);
      acc.z += (q0.z + q1.z) + (q2.z + q3.z); acc.w += (q0.w + q1.w) + (q2.w + q3.w);
    }
    for (; c < a.nchunks; c += S) {
      const float4 q = src[(size_t)c * cstride];
      acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
    }
  }
  if (slice > 0) sPart[(slice - 1) * per + lane] = acc;
  __syncthreads();
  if (slice == 0 && v < nvec) {
    for (int k = 0; k < S - 1; ++k) { const float4 q = sPart[k * per + lane]; acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
    const int ol = (int)(v % ob4) * 4, cl = (int)((v / ob4) % CB), tap = (int)((v / ((size_t)ob4 * CB)) % a.ntaps);
    const int blk = (int)(v / ((size_t)ob4 * CB * a.ntaps));
    const int ci = (blk / a.oblocks) * CB + cl, co = (blk % a.oblocks) * OB + ol;
    if (ci < a.Cin) {
      float* dst = a.dw + ((size_t)tap * a.Cin + ci) * a.Cout + co;
      if ((a.Cout & 3) == 0 && co + 3 < a.Cout) {      // one 16-byte read-modify-write (co is a multiple of 4, dW 16-byte aligned)
        float4 t = *reinterpret_cast<float4*>(dst);
        t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
        *reinterpret_cast<float4*>(dst) = t;
      } else {
        const float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (co + k < a.Cout) dst[k] += r[k];
      }
    }
  }
  if (rb == 0 && a.db != nullptr) {   // bias gradient of this job: [chunk][co block][OB] partials, tiny
    // 16 output channels x 16 chunk slices at a time (slice q: chunks q, q+16, ... in order; then the 16 slice sums in order).
    // One thread per channel walking all chunks was a chain of up to 256 dependent-latency loads in ONE block: 20-27 us of a
    // launch whose other blocks finish in ~5
    __syncthreads();                  // sPart is free again
    float* sB = reinterpret_cast<float*>(sPart);
    const int bl = threadIdx.x & 15, bs = threadIdx.x >> 4;
    for (int co0 = 0; co0 < a.Cout; co0 += 16) {
      const int co = co0 + bl;
      float t = 0.f;
      if (co < a.Cout) {
        const float* src = a.ws_db + (size_t)(co / OB) * OB + co % OB;
        const size_t cs = (size_t)a.oblocks * OB;
        int c = bs;
        for (; c + 48 < a.nchunks; c += 64) {
          const float q0 = src[(size_t)c * cs], q1 = src[(size_t)(c + 16) * cs], q2 = src[(size_t)(c + 32) * cs], q3 = src[(size_t)(c + 48) * cs];
          t += (q0 + q1) + (q2 + q3);
        }
        for (; c < a.nchunks; c += 16) t += src[(size_t)c * cs];
      }
      sB[threadIdx.x] = t;
      __syncthreads();
      if (bs == 0 && co < a.Cout) {
        float s2 = 0.f;
        for (int k = 0; k < 16; ++k) s2 += sB[k * 16 + bl];
        a.db[co] += s2;
      }
      __syncthreads();
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------
struct Geo { int tpw, cbf, obf, tw, nw, cs, os, xr, bm; bool narrow, precise; int up; };

// hdrsky_conv2d_wgrad_kernel_names: while a planning pass runs with this pointer set, every launch it WOULD make appends
// its kernel's name as rocprofv3 prints it (bench.py labels its roofline rows with them)
static thread_local std::string* g_wg_names = nullptr;
static void note_kernel(const char* fmt, ...) {
  if (!g_wg_names) return;
  char buf[192];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (!g_wg_names->empty()) *g_wg_names += " + ";
  *g_wg_names += buf;
}

template <int TPW, int CBF, int OBF, int TW, bool NARROW, bool PRECISE, int UP, int NW, int CS, int OS, int XR, int BM>
struct WgradVariant {
  static constexpr int NT = NW * 64, CB = CBF * 16, OB = OBF * 16, TH = BM / TW, NTG = NW / (CS * OS);
  // fills the geometry-dependent fields of one job; returns the LDS bytes it needs, or a negative error
  static int prepare(WgradArgs& a, int wg_target) {
    if (UP != (a.da_k > 0 ? 2 : (a.upsample == 2 ? 1 : 0))) return HDRSKY_EUNSUPPORTED;
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
    if (cdiv(a.ntaps, NTG) > TPW) return HDRSKY_EUNSUPPORTED;
    if (!NARROW && (a.Cin % CB) != 0) return HDRSKY_EUNSUPPORTED;   // the X staging reads whole CB-channel blocks
    if (cdiv(a.NPIX * (CB / 8), NT) > (UP ? XR / 4 : XR)) return HDRSKY_EUNSUPPORTED;   // register prefetch budget
    a.RX = CB * 2 + 16;
    a.RY = OB * 2 + 16;
    const int planes = PRECISE ? 2 : 1;
    const int xbytes = roundup(a.NPIX * a.RX, 16), ybytes = roundup(BM * a.RY, 16);
    a.off_xlo = xbytes;
    a.off_y = xbytes * planes;
    a.off_ylo = a.off_y + ybytes;
    a.off_ss = a.off_y + ybytes * planes;
    // pixel split: enough workgroups to fill the chip, few enough that the atomic traffic (one dW block per
    // workgroup) stays small
    const int nblk = a.cblocks * a.oblocks;
    int chunks = cdiv(wg_target, nblk);
    if (chunks < 1) chunks = 1;
    if (chunks > a.ntiles) chunks = a.ntiles;
    a.tiles_per_wg = cdiv(a.ntiles, chunks);
    a.nchunks = cdiv(a.ntiles, a.tiles_per_wg);
    const int tps = a.tiles_x * a.tiles_y;
    const int nsamp = (a.tiles_per_wg + tps - 2) / tps + 1;          // samples a run of tiles_per_wg tiles can touch
    a.off_red = a.off_ss + nsamp * 2 * CB * 4;
    a.off_da = a.off_red + NT * 8 * 4;
    if (UP == 2 && (a.da_C % CB) != 0) return HDRSKY_EUNSUPPORTED;   // one tap per block of virtual channels
    const int lds = a.off_da + (UP == 2 ? a.H * 2 * 4 : 0);
    if (lds > 160 * 1024) return HDRSKY_EUNSUPPORTED;
    return lds;
  }
  static void note() {
    note_kernel("conv_wgrad_kernel<%d, %d, %d, %d, %s, %s, %d, %d, %d, %d, %d, %d>", TPW, CBF, OBF, TW, NARROW ? "true" : "false",
                PRECISE ? "true" : "false", UP, NW, CS, OS, XR, BM);
  }
  static int launch(MultiArgs& m, int lds, hipStream_t stream) {
    auto kern = conv_wgrad_kernel<TPW, CBF, OBF, TW, NARROW, PRECISE, UP, NW, CS, OS, XR, BM>;
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024) != hipSuccess)
        return HDRSKY_ELAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(m.first[m.njobs]), dim3(NT), lds, stream, m);
    HDRSKY_CHECK_LAUNCH();
    return HDRSKY_OK;
  }
};

// Runs `fn` on the kernel variant that implements geometry `g` (the instantiated set).
template <typename F>
int with_variant(const Geo& g, F&& fn) {
#define HDRSKY_WG_(TPW_, CBF_, OBF_, NARROW_, UP_, NW_, CS_, OS_, XR_, BM_, TW_)                                     \
  if (g.tpw == TPW_ && g.cbf == CBF_ && g.obf == OBF_ && g.narrow == NARROW_ && g.up == UP_ && g.nw == NW_ &&        \
      g.bm == BM_ && g.tw == TW_) {                                                                                   \
    if (g.precise) return fn(WgradVariant<TPW_, CBF_, OBF_, TW_, NARROW_, true, UP_, NW_, CS_, OS_, XR_, BM_>());     \
    return fn(WgradVariant<TPW_, CBF_, OBF_, TW_, NARROW_, false, UP_, NW_, CS_, OS_, XR_, BM_>());                   \
  }
#define HDRSKY_WG(TPW_, CBF_, OBF_, NARROW_, UP_, NW_, CS_, OS_, XR_)                                                \
  HDRSKY_WG_(TPW_, CBF_, OBF_, NARROW_, UP_, NW_, CS_, OS_, XR_, 128, 32)                                            \
  HDRSKY_WG_(TPW_, CBF_, OBF_, NARROW_, UP_, NW_, CS_, OS_, XR_, 128, 16)
  // 16 waves, 32x32-channel dW blocks (a layer on its own)
  HDRSKY_WG(3, 2, 2, false, 0, 16, 2, 2, 3) HDRSKY_WG(3, 2, 2, false, 1, 16, 2, 2, 4)
  HDRSKY_WG(4, 2, 2, false, 0, 16, 2, 2, 3) HDRSKY_WG(7, 2, 1, false, 0, 16, 2, 1, 2)
  HDRSKY_WG(2, 1, 4, true, 0, 16, 1, 2, 3) HDRSKY_WG(2, 1, 2, true, 0, 16, 1, 2, 3)
  HDRSKY_WG(7, 1, 2, true, 0, 16, 1, 2, 2)
  // 64-pixel tiles for outputs of at most four 16-pixel rows (the 4x16 maps of the discriminator / sun-radiance stacks)
  HDRSKY_WG_(4, 2, 2, false, 0, 16, 2, 2, 2, 64, 16) HDRSKY_WG_(3, 2, 2, false, 0, 16, 2, 2, 3, 64, 16)
  // 8 waves, 64x64-channel dW blocks (several wide layers in one launch)
  HDRSKY_WG(3, 4, 4, false, 0, 8, 2, 1, 4)
  // distortion-aware layers (1x1 over k*k*C virtual channels, every wave on the one tap): 64x64 blocks when C % 64 == 0,
  // else 32x32
  HDRSKY_WG(1, 4, 4, false, 2, 8, 4, 2, 8) HDRSKY_WG(1, 2, 2, false, 2, 4, 2, 2, 8)
#undef HDRSKY_WG
#undef HDRSKY_WG_
  return HDRSKY_EUNSUPPORTED;
}

// Geometry for one layer: `big` = 64x64-channel blocks (grouped launches of wide stride-1 layers).
static Geo choose_geo(const hdrsky_wgrad_job& job, bool big) {
  const hdrsky_conv_desc* d = &job.desc;
  Geo g{};
  const int ntaps = d->KH * d->KW;
  g.narrow = d->Cin <= 8;
  g.precise = d->compute == HDRSKY_BF16X3;
  g.up = d->upsample == 2 ? 1 : 0;
  g.tw = d->Wo >= 32 ? 32 : 16;
  g.bm = 128;
  if (job.da_ksize > 0) {     // distortion-aware layer: 1x1 over k*k*C virtual channels
    g.up = 2; g.tpw = 1; g.narrow = false;
    if (job.da_C % 64 == 0) { g.nw = 8; g.cs = 4; g.os = 2; g.cbf = 4; g.obf = 4; }
    else { g.nw = 4; g.cs = 2; g.os = 2; g.cbf = 2; g.obf = 2; }
    return g;
  }
  if (big) {
    g.nw = 8; g.cs = 2; g.os = 1; g.cbf = 4; g.obf = 4; g.tpw = ntaps <= 12 ? 3 : 4;
    return g;
  }
  g.nw = 16;
  if (g.narrow) { g.cbf = 1; g.obf = (d->Cout >= 64 && ntaps <= 16) ? 4 : 2; g.tpw = ntaps <= 16 ? 2 : 7; }
  else if (ntaps > 16) { g.cbf = 2; g.obf = 1; g.tpw = 7; }
  else {
    g.cbf = 2; g.obf = 2; g.tpw = ntaps <= 12 ? 3 : 4;
    if (g.tw == 16 && d->Ho <= 4 && !g.up) g.bm = 64;   // a 128-pixel tile would be half padding
  }
  return g;
}

static bool can_go_big(const hdrsky_wgrad_job& job) {
  const hdrsky_conv_desc* d = &job.desc;
  return job.da_ksize == 0 && d->Cin >= 64 && d->Cout >= 64 && (d->Cin % 64) == 0 && d->stride == 1 && d->upsample == 1 && d->KH * d->KW <= 9;
}

static int fill_job(WgradArgs& a, const hdrsky_wgrad_job& j) {
  const hdrsky_conv_desc* d = &j.desc;
  if (!j.x || !j.dy || !j.dw) return HDRSKY_EINVAL;
  if (d->dilate != 1) return HDRSKY_EUNSUPPORTED;
  const bool narrow = d->Cin <= 8;
  if (!narrow && (d->Cin % 32) != 0) return HDRSKY_EUNSUPPORTED;
  if (narrow && d->upsample != 1) return HDRSKY_EUNSUPPORTED;
  if (d->in_mode == HDRSKY_IN_AFFINE && (!j.in_scale || !j.in_shift)) return HDRSKY_EINVAL;
  if (d->in_mode == HDRSKY_IN_PARTIALS && (!j.in_part || !j.in_gamma || !j.in_beta || d->in_nparts <= 0)) return HDRSKY_EINVAL;
  a = WgradArgs{};
  a.x = j.x; a.dy = j.dy; a.dw = j.dw; a.db = j.db;
  a.in_scale = j.in_scale; a.in_shift = j.in_shift; a.in_part = j.in_part; a.in_gamma = j.in_gamma; a.in_beta = j.in_beta;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.upsample = d->upsample; a.Hc = d->Hc; a.Wc = d->Wc;
  a.in_mode = d->in_mode; a.ss_bstride = d->ss_bstride; a.in_nparts = d->in_nparts;
  a.in_eps = d->in_eps; a.in_inv_count = 1.f / (float)(d->H * d->W); a.in_slope = d->in_slope;
  a.x_bf16 = j.x_bf16; a.dy_bf16 = j.dy_bf16;
  // (a bf16 x with an operand transform is a raw conv output stored as bf16: widened into the staging registers, the
  // transform applied there as for fp32 storage)
  if (j.x_bf16 && (narrow || d->upsample != 1)) return HDRSKY_EUNSUPPORTED;
  if (j.dy_bf16 && (d->Cout & 7) != 0) return HDRSKY_EUNSUPPORTED;
  if (j.da_ksize > 0) {
    // desc describes the 1x1 weight gradient over the virtual channels; x is the layer's real input [B,H,W,da_C] fp32
    const int k = j.da_ksize;
    if (!j.da_offs || (k & 1) == 0 || k > 7 || j.da_C <= 0 || (j.da_C % 32) != 0 || d->Cin != k * k * j.da_C) return HDRSKY_EINVAL;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->upsample != 1 || d->H != d->Ho || d->W != d->Wo) return HDRSKY_EINVAL;
    if (d->in_mode != HDRSKY_IN_NONE || d->in_slope != 1.f || j.x_bf16) return HDRSKY_EUNSUPPORTED;
    a.da_offs = j.da_offs; a.da_k = k; a.da_C = j.da_C;
  }
  return HDRSKY_OK;
}

static bool same_geo(const Geo& p, const Geo& q) {
  return p.tpw == q.tpw && p.cbf == q.cbf && p.obf == q.obf && p.tw == q.tw && p.nw == q.nw && p.bm == q.bm && p.narrow == q.narrow &&
         p.precise == q.precise && p.up == q.up;
}


// The fixed-order reduce of the LDS-DMA and the narrow-layer kernels' partial slabs: ONE launch per call of
// hdrsky_conv2d_wgrad_multi_det for the jobs of all their compute launches (each used to be followed by its own: four launches
// for a call with both kinds of layer, the two reduces ~10 us each of mostly launch latency), block shape and slice count per job.
struct PendingReduce {
  MultiArgs mr{};
  int rblocks = 0;
  bool plan_only = false;
  void* stream = nullptr;
  int flush() {
    if (mr.njobs > 0 && rblocks > 0) {
      mr.rfirst[mr.njobs] = rblocks;
      if (plan_only) note_kernel("wgrad_reduce_kernel");
      else {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, (hipStream_t)stream, mr, 0, 0, 0);
        HDRSKY_CHECK_LAUNCH();
      }
    }
    mr = MultiArgs{}; rblocks = 0;
    return HDRSKY_OK;
  }
  // appends one job's reduce (nblocks4 = float4 elements of one chunk's slab set); flushes first when the argument block is full
  int add(const WgradArgs& ar, size_t nvec, int S) {
    if (mr.njobs == WG_MAXJ) { const int rc = flush(); if (rc != HDRSKY_OK) return rc; }
    WgradArgs& d = mr.job[mr.njobs];
    d = ar; d.rS = S;
    mr.rfirst[mr.njobs] = rblocks;
    rblocks += (int)((nvec + 256 / S - 1) / (256 / S));
    ++mr.njobs;
    return HDRSKY_OK;
  }
};

// ---- v2 host side -------------------------------------------------------------------------------------------------------
static bool v2_eligible(const hdrsky_wgrad_job& j) {
  const hdrsky_conv_desc* d = &j.desc;
  const int s2min = hdrsky_hooks().wgrad2_s2min;   // (tuning hook; 32)
  return j.x_bf16 && j.dy_bf16 && j.da_ksize == 0 && d->compute == HDRSKY_BF16 && d->upsample == 1 && d->dilate == 1 &&
         // (stride 2: a tile's input patch is ~4x its output and every pixel of it is copied - such layers run on 64-pixel
         // tiles (wg2_prepare); measured, batch 32: 1.25-1.8x the register-staged kernel, 1.9-2.5x at 128x512)
         (d->stride == 1 || (d->stride == 2 && d->Cin >= s2min)) && d->Cin >= 32 && d->Cin <= 4096 && (d->Cin % 32) == 0 && d->Cout >= 32 && (d->Cout % 32) == 0 &&
         d->in_mode == HDRSKY_IN_NONE && d->in_slope == 1.f && j.x && j.dy && j.dw && d->KH * d->KW <= 64;
}

constexpr int WG2_UPW = 5;      // units per compute wave (80 accumulator registers at 4 co fragments)

// fills a job's geometry; returns the LDS bytes of its launch, or a negative error
static int wg2_prepare(Wg2Args& a, const hdrsky_wgrad_job& j, int wg_target) {
  const hdrsky_conv_desc* d = &j.desc;
  a = Wg2Args{};
  a.x = (const unsigned short*)j.x; a.dy = (const unsigned short*)j.dy;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.ntaps = d->KH * d->KW;
  a.cbf = (a.Cin % 64) == 0 ? 4 : 2;
  a.obf = (a.Cout % 64) == 0 ? 4 : 2;
  const int CB = a.cbf * 16, OB = a.obf * 16, SPX = CB / 8 + 2, SPY = OB / 8 + 2;
  const int TW = d->Wo >= 32 ? 32 : 16;
  a.tw_shift = TW == 32 ? 5 : 4;
  // 64-pixel tiles for 4-row maps (a 128-pixel tile would be half padding) and for stride 2 (a tile's input patch is ~4x its
  // output: 128 output pixels of a 4x4 stride-2 layer need 84 KB per stage)
  a.BM = ((TW == 16 && d->Ho <= 4) || d->stride == 2) ? 64 : 128;
  a.TH = a.BM / TW;
  a.tiles_x = cdiv(a.Wo, TW); a.tiles_y = cdiv(a.Ho, a.TH);
  a.ntiles = a.B * a.tiles_x * a.tiles_y;
  a.cblocks = a.Cin / CB; a.oblocks = a.Cout / OB;
  const int maxt = WG2_UPW * 8 / a.cbf;
  if (a.ntaps <= maxt) a.tgs = a.ntaps;
  else { const int rows = maxt / a.KW; if (rows < 1) return HDRSKY_EUNSUPPORTED; a.tgs = rows * a.KW; }
  a.ntg = cdiv(a.ntaps, a.tgs);
  a.WT = (TW - 1) * a.stride + a.KW;
  a.wt_magic = ((1 << 24) + a.WT - 1) / a.WT;
  const int rows_g = cdiv(a.tgs, a.KW) < a.KH ? cdiv(a.tgs, a.KW) : a.KH;
  const int npix = ((a.TH - 1) * a.stride + rows_g) * a.WT;
  if ((long)npix * SPX * a.wt_magic >= (1L << 32)) return HDRSKY_EUNSUPPORTED;
  a.xbytes = roundup(npix * SPX * 16, 1024);
  const int ybytes = roundup(a.BM * SPY * 16, 1024);
  a.stage_bytes = a.xbytes + ybytes;
  a.ns = 3 * a.stage_bytes <= 160 * 1024 ? 3 : (2 * a.stage_bytes <= 160 * 1024 ? 2 : 0);
  if (a.ns == 0) return HDRSKY_EUNSUPPORTED;
  if (a.xbytes > 16 * 4 * 1024 || ybytes > 6 * 4 * 1024) return HDRSKY_EUNSUPPORTED;   // slot registers: 16 + 6 pieces per loader wave
  if (a.WT >= 1024 || (a.TH - 1) * a.stride + rows_g >= 1024) return HDRSKY_EUNSUPPORTED;
  // pixel split: what the launch's workgroup budget allows, but (a) at least two tiles per workgroup (a one-tile workgroup
  // pipelines nothing; eight - the first rule - left the encoder's stride-2 layers on 96 workgroups: 35 -> 30 us alone, step
  // -0.5 %; every other call is bounded by its work share first) and
  // (b) at most ~34 MB of partial slabs per layer for the reduce launch to read back (measured, batch 32: 3x3 32->64 at
  // 16x64 split 256 ways = 15 us + 57 us of reduce; 3x3 128->128 at 8x32 split 64 ways = 17 + 21 us)
  const int base = a.cblocks * a.oblocks * a.ntg;
  int chunks = (wg_target + base / 2) / base;
  const long dw_bytes = (long)a.ntaps * a.Cin * a.Cout * 4;
  const int mint = hdrsky_hooks().wgrad2_mint;   // (tuning hook; 2)
  if (chunks > a.ntiles / mint) chunks = a.ntiles / mint;
  if ((long)chunks * dw_bytes > (34L << 20)) chunks = (int)((34L << 20) / dw_bytes);
  if (chunks < 1) chunks = 1;
  if (chunks > a.ntiles) chunks = a.ntiles;
  a.tiles_per_wg = cdiv(a.ntiles, chunks);
  a.nchunks = cdiv(a.ntiles, a.tiles_per_wg);
  return a.ns * a.stage_bytes;
}

static int wg2_launch(Multi2Args& m, int lds, hipStream_t stream) {
  auto kern = conv_wgrad2_kernel<WG2_UPW>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return HDRSKY_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(m.first[m.njobs]), dim3(768), lds, stream, m);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

static unsigned long long* g_wg2_stamps = nullptr;
extern "C" void hdrsky_debug_wgrad2_stamps(void* buf) { g_wg2_stamps = (unsigned long long*)buf; }

// Launches (or, plan_only, sizes) the v2 kernel + the shared reduce for the eligible jobs among jobs[0..njobs); marks them
// in done[].  Jobs the kernel cannot take (LDS budget, geometry) stay unmarked: the caller's v1 path handles them.
static int wgrad2_groups(const hdrsky_wgrad_job* jobs, int njobs, bool* done, float* ws, size_t ws_floats, size_t* ws_used,
                         bool plan_only, void* stream, PendingReduce& red) {
  const int wg_hook = hdrsky_hooks().wgrad2_wgs;   // (tuning hook; 0 = by work share)
  // 192 workgroups of 768 threads, not one per compute unit: inside the three-stream step the quarter of the chip the launch leaves
  // free is worth more to the other streams than the shorter launch is to this one, and fewer pixel chunks mean smaller partial
  // slabs for the reduce (round 5: 32x128 step -1.2 ... -1.7 %, 128x512 step -0.4 ... -1.3 % against 256; 128 and 320+ are worse:
  // profiles/r05_wgrad_wgs_ab*.txt)
  const int wg_total = wg_hook > 0 ? wg_hook : 192;
  int members[256], nm = 0;
  double work[256];
  for (int k = 0; k < njobs; ++k) {
    if (done[k] || !v2_eligible(jobs[k])) continue;
    Wg2Args probe;
    if (wg2_prepare(probe, jobs[k], 1) < 0) continue;
    const hdrsky_conv_desc& d = jobs[k].desc;
    work[nm] = (double)d.B * d.Ho * d.Wo * d.KH * d.KW * d.Cin * d.Cout;
    members[nm++] = k;
  }
  for (int base = 0; base < nm; base += WG2_MAXJ) {
    const int cnt = nm - base < WG2_MAXJ ? nm - base : WG2_MAXJ;
    double wpart = 0.0;
    for (int q = 0; q < cnt; ++q) wpart += work[base + q];
    Multi2Args m2{};
    m2.nt = hdrsky_hooks().wgrad2_nt;
    m2.njobs = cnt;
    m2.stamps = g_wg2_stamps;
    int lds = 0, blocks = 0, maxchunks = 1;
    WgradArgs rjobs[WG2_MAXJ]; size_t rvec[WG2_MAXJ];
    for (int q = 0; q < cnt; ++q) {
      Wg2Args tmp;
      if (wg2_prepare(tmp, jobs[members[base + q]], (int)(wg_total * work[base + q] / wpart + 0.5)) >= 0 && tmp.nchunks > maxchunks)
        maxchunks = tmp.nchunks;
    }
    const int S = maxchunks >= 32 ? 16 : 4;
    for (int q = 0; q < cnt; ++q) {
      const hdrsky_wgrad_job& j = jobs[members[base + q]];
      Wg2Args& a = m2.job[q];
      const int r = wg2_prepare(a, j, (int)(wg_total * work[base + q] / wpart + 0.5));
      if (r < 0) return r;
      if (r > lds) lds = r;
      const int CB = a.cbf * 16, OB = a.obf * 16;
      m2.first[q] = blocks;
      blocks += a.cblocks * a.oblocks * a.ntg * a.nchunks;
      a.direct = a.nchunks == 1 ? 1 : 0;
      a.dw = j.dw; a.db = j.db;
      const size_t nslab = a.direct ? 0 : (size_t)a.nchunks * a.cblocks * a.oblocks * a.ntaps * CB * OB;
      const size_t nbias = a.direct ? 0 : (size_t)a.nchunks * a.oblocks * OB;
      a.ws = ws + *ws_used;
      a.ws_db = (j.db != nullptr && !a.direct) ? ws + *ws_used + nslab : nullptr;
      *ws_used += nslab + (j.db != nullptr ? nbias : 0);
      WgradArgs& ar = rjobs[q];      // what wgrad_reduce_kernel reads
      ar = WgradArgs{};
      ar.dw = j.dw; ar.db = j.db; ar.ws = a.ws; ar.ws_db = a.ws_db; ar.Cin = a.Cin; ar.Cout = a.Cout;
      ar.nchunks = a.nchunks; ar.cblocks = a.cblocks; ar.oblocks = a.oblocks; ar.ntaps = a.ntaps;
      ar.RX = CB; ar.RY = OB;        // per-job block shape of the reduce launch (CB = OB = 0 arguments)
      rvec[q] = a.direct ? 0 : (size_t)a.cblocks * a.oblocks * a.ntaps * CB * (OB / 4);
    }
    m2.first[cnt] = blocks;
    if (plan_only) note_kernel("conv_wgrad2_kernel<%d>", WG2_UPW);
    else {
      if (*ws_used > ws_floats) return HDRSKY_EINVAL;
      const int r = wg2_launch(m2, lds, (hipStream_t)stream);
      if (r != HDRSKY_OK) return r;
    }
    for (int q = 0; q < cnt; ++q)
      if (rvec[q] > 0) { const int rc = red.add(rjobs[q], rvec[q], S); if (rc != HDRSKY_OK) return rc; }
    for (int q = 0; q < cnt; ++q) done[members[base + q]] = true;
  }
  return HDRSKY_OK;
}

// ---- v3 host side -------------------------------------------------------------------------------------------------------
constexpr int WG3_UPW = 2;

// 0: not for conv_wgrad3_kernel, 1: narrow input (Cin <= 8), 2: narrow output (Cout <= 4)
static int wg3_kind(const hdrsky_wgrad_job& j) {
  const hdrsky_conv_desc* d = &j.desc;
  if (j.da_ksize != 0 || d->compute != HDRSKY_BF16 || d->upsample != 1 || d->dilate != 1 || !j.x || !j.dy || !j.dw) return 0;
  if (d->Cin <= 8 && d->Cout >= 16 && (d->Cout % 16) == 0 && !j.x_bf16 && d->in_mode == HDRSKY_IN_NONE && d->in_slope == 1.f &&
      (d->stride == 1 || d->stride == 2))
    return 1;
  if (d->Cout <= 4 && d->Cin >= 16 && (d->Cin % 16) == 0 && d->stride == 1 && !j.dy_bf16) {
    // (x_bf16 with a transform: a raw conv output stored as bf16 - widened and transformed while staging, as fp32 storage is)
    if (d->in_mode == HDRSKY_IN_AFFINE && (!j.in_scale || !j.in_shift)) return 0;
    if (d->in_mode == HDRSKY_IN_PARTIALS && (!j.in_part || !j.in_gamma || !j.in_beta || d->in_nparts <= 0)) return 0;
    return 2;
  }
  return 0;
}

// fills a job's geometry (wg_target workgroups, if the layer has that many 4-tile chunks); returns its LDS bytes or an error
static int wg3_prepare(Wg3Args& a, const hdrsky_wgrad_job& j, int wg_target) {
  const hdrsky_conv_desc* d = &j.desc;
  const int kind = wg3_kind(j);
  if (kind == 0) return HDRSKY_EUNSUPPORTED;
  a = Wg3Args{};
  a.B = d->B; a.KH = d->KH; a.KW = d->KW;
  if (kind == 1) {
    a.wide = j.dy; a.wide_bf16 = j.dy_bf16; a.Hk = d->Ho; a.Wk = d->Wo; a.Cw = d->Cout;
    a.narrow = j.x; a.Hn = d->H; a.Wn = d->W; a.Cn = d->Cin;
    a.stride = d->stride; a.PT = d->pad_t; a.PL = d->pad_l; a.transposed = 0; a.bias_mode = j.db ? 1 : 0;
    a.in_mode = HDRSKY_IN_NONE; a.in_slope = 1.f;
  } else {
    a.wide = j.x; a.wide_bf16 = j.x_bf16; a.Hk = d->H; a.Wk = d->W; a.Cw = d->Cin;
    a.narrow = j.dy; a.Hn = d->Ho; a.Wn = d->Wo; a.Cn = d->Cout;
    a.stride = 1; a.PT = d->KH - 1 - d->pad_t; a.PL = d->KW - 1 - d->pad_l; a.transposed = 1; a.bias_mode = j.db ? 2 : 0;
    a.in_scale = j.in_scale; a.in_shift = j.in_shift; a.in_part = j.in_part; a.in_gamma = j.in_gamma; a.in_beta = j.in_beta;
    a.in_mode = d->in_mode; a.ss_bstride = d->ss_bstride; a.in_nparts = d->in_nparts; a.in_eps = d->in_eps;
    a.in_inv_count = 1.f / (float)(d->H * d->W); a.in_slope = d->in_slope;
  }
  a.CP = a.Cn <= 4 ? 4 : 8;
  a.XPF = 16 / a.CP;
  a.NFX = cdiv(a.KW, a.XPF);
  a.NMF = a.KH * a.NFX;
  a.NB = (a.Cw % 64) == 0 ? 64 : ((a.Cw % 32) == 0 ? 32 : 16);
  a.NF = a.NB / 16;
  a.nblocks = a.Cw / a.NB;
  a.nsplit = (a.NMF <= 4 && a.NF >= 2) ? 2 : 1;
  a.nunits = a.NMF * a.nsplit;
  if (cdiv(a.nunits, 8) > WG3_UPW) return HDRSKY_EUNSUPPORTED;
  const int TW = a.Wk >= 32 ? 32 : 16;
  a.tw_shift = TW == 32 ? 5 : 4;
  // 128-pixel tiles, or 64 when that is what lets a thread hold ONE wide item and ONE narrow pixel per tile (the D = 4
  // instantiations: four tiles' loads in flight) - the stride-2 layers, 64-channel blocks - and for 4-row maps
  for (a.BM = (TW == 16 && a.Hk <= 4) ? 64 : 128;; a.BM = 64) {
    a.TH = a.BM / TW;
    a.HT = (a.TH - 1) * a.stride + a.KH;
    a.WTn = (TW - 1) * a.stride + a.NFX * a.XPF;
    a.npixn = a.HT * a.WTn;
    a.deep = (cdiv(a.npixn, 512) <= 1 && cdiv(a.BM * a.NB / 8, 512) <= 1) ? 1 : 0;
    if (a.deep || a.BM == 64) break;
  }
  if (!a.deep) {                                          // the 2-deep instantiation: up to two of each, 128-pixel tiles
    a.BM = (TW == 16 && a.Hk <= 4) ? 64 : 128;
    a.TH = a.BM / TW;
    a.HT = (a.TH - 1) * a.stride + a.KH;
    a.npixn = a.HT * a.WTn;
    if (cdiv(a.npixn, 512) > 2 || cdiv(a.BM * a.NB / 8, 512) > 2) return HDRSKY_EUNSUPPORTED;
  }
  if (a.HT >= 256) return HDRSKY_EUNSUPPORTED;
  a.tiles_x = cdiv(a.Wk, TW); a.tiles_y = cdiv(a.Hk, a.TH);
  a.ntiles = a.B * a.tiles_x * a.tiles_y;
  a.wt_magic = ((1 << 24) + a.WTn - 1) / a.WTn;
  a.RY = (a.NB / 8 + 2) * 16;
  a.nbytes = roundup(a.npixn * a.CP * 2, 16);
  a.stage_bytes = a.nbytes + a.BM * a.RY;
  a.off_ss = 2 * a.stage_bytes;
  if (a.off_ss < 512 * 8 * 4) a.off_ss = 512 * 8 * 4;   // the bias reduction's scratch overlays the stages
  int chunks = wg_target / a.nblocks;
  const int minpx = hdrsky_hooks().wgrad3_minpx;   // (tuning hook; 256)
  // at least this many pixels per workgroup (512: the 4x4 stride-2 first layers ran on 64 workgroups - 26.7 -> 21.9 us at 256,
  // the 512->1 head 18.6 -> 14.0; 128 is no better)
  const int mint = minpx / a.BM > 0 ? minpx / a.BM : 1;
  if (chunks > a.ntiles / mint) chunks = a.ntiles / mint;
  if (chunks < 1) chunks = 1;
  a.tiles_per_wg = cdiv(a.ntiles, chunks);
  a.nchunks = cdiv(a.ntiles, a.tiles_per_wg);
  const int tps = a.tiles_x * a.tiles_y;
  a.nsamp = (a.tiles_per_wg + tps - 2) / tps + 1;
  if (kind == 1) { a.slabCB = a.CP; a.slabOB = a.NB; } else { a.slabCB = a.NB; a.slabOB = 4; }
  const int lds = a.off_ss + a.nsamp * 2 * a.NB * 4;
  if (lds > 80 * 1024) return HDRSKY_EUNSUPPORTED;       // two workgroups per CU
  return lds;
}

// variant: 0 = D 4, four registers per narrow pixel; 1 = D 4, eight; 2 = D 2, two wide items + two narrow pixels per thread
template <int D, int WI, int NI, int CPR>
static int wg3_launch_as(Multi3Args& m, int lds, hipStream_t stream) {
  auto kern = conv_wgrad3_kernel<D, WI, NI, CPR, WG3_UPW>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
      return HDRSKY_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(m.first[m.njobs]), dim3(512), lds, stream, m);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}
static int wg3_launch(int variant, Multi3Args& m, int lds, hipStream_t stream) {
  if (variant == 0) return wg3_launch_as<4, 1, 1, 4>(m, lds, stream);
  if (variant == 1) return wg3_launch_as<4, 1, 1, 8>(m, lds, stream);
  return wg3_launch_as<2, 2, 2, 8>(m, lds, stream);
}

// Launches (or, plan_only, sizes) the narrow-layer kernel + the shared reduce for the jobs it takes; marks them in done[].
static int wgrad3_groups(const hdrsky_wgrad_job* jobs, int njobs, bool* done, float* ws, size_t ws_floats, size_t* ws_used,
                         bool plan_only, void* stream, PendingReduce& red) {
  const int wg_each = hdrsky_hooks().wgrad3_wgs;   // (tuning hook: workgroups per layer; 256)
  int all[256], na = 0, cls[256];
  bool deep8 = false;
  for (int k = 0; k < njobs; ++k) {
    if (done[k]) continue;
    Wg3Args probe;
    if (wg3_prepare(probe, jobs[k], wg_each) < 0) continue;
    cls[na] = probe.deep ? 0 : 2;
    if (probe.deep && probe.CP == 8) deep8 = true;
    all[na++] = k;
  }
  for (int pass = 0; pass < 2; ++pass) {                  // the deep jobs of a call share a launch, the others another
  int members[256], nm = 0;
  for (int q = 0; q < na; ++q) if ((cls[q] == 2) == (pass == 1)) members[nm++] = all[q];
  const int variant = pass == 1 ? 2 : (deep8 ? 1 : 0);
  for (int base = 0; base < nm; base += WG3_MAXJ) {
    const int cnt = nm - base < WG3_MAXJ ? nm - base : WG3_MAXJ;
    Multi3Args m3{};
    m3.njobs = cnt;
    m3.stamps = g_wg2_stamps;
    int lds = 0, blocks = 0, maxchunks = 1;
    WgradArgs rjobs[WG3_MAXJ]; size_t rvec[WG3_MAXJ];
    for (int q = 0; q < cnt; ++q) {
      Wg3Args tmp;
      if (wg3_prepare(tmp, jobs[members[base + q]], wg_each) >= 0 && tmp.nchunks > maxchunks) maxchunks = tmp.nchunks;
    }
    const int S = maxchunks >= 32 ? 16 : 4;
    for (int q = 0; q < cnt; ++q) {
      const hdrsky_wgrad_job& j = jobs[members[base + q]];
      Wg3Args& a = m3.job[q];
      const int r = wg3_prepare(a, j, wg_each);
      if (r < 0) return r;
      if (r > lds) lds = r;
      m3.first[q] = blocks;
      blocks += a.nblocks * a.nchunks;
      const int ntaps = a.KH * a.KW;
      const size_t nslab = (size_t)a.nchunks * a.nblocks * ntaps * a.slabCB * a.slabOB;
      const int oblocks = a.transposed ? 1 : a.nblocks, cblocks = a.transposed ? a.nblocks : 1;
      const size_t nbias = (size_t)a.nchunks * oblocks * a.slabOB;
      a.ws = ws + *ws_used;
      a.ws_db = j.db != nullptr ? ws + *ws_used + nslab : nullptr;
      *ws_used += nslab + (j.db != nullptr ? nbias : 0);
      WgradArgs& ar = rjobs[q];      // what wgrad_reduce_kernel reads
      ar = WgradArgs{};
      ar.dw = j.dw; ar.db = j.db; ar.ws = a.ws; ar.ws_db = a.ws_db; ar.Cin = j.desc.Cin; ar.Cout = j.desc.Cout;
      ar.nchunks = a.nchunks; ar.cblocks = cblocks; ar.oblocks = oblocks; ar.ntaps = ntaps;
      ar.RX = a.slabCB; ar.RY = a.slabOB;
      rvec[q] = (size_t)a.nblocks * ntaps * a.slabCB * (a.slabOB / 4);
    }
    m3.first[cnt] = blocks;
    if (plan_only)
      note_kernel(variant == 0 ? "conv_wgrad3_kernel<4, 1, 1, 4, %d>" : variant == 1 ? "conv_wgrad3_kernel<4, 1, 1, 8, %d>" : "conv_wgrad3_kernel<2, 2, 2, 8, %d>", WG3_UPW);
    else {
      if (*ws_used > ws_floats) return HDRSKY_EINVAL;
      const int r = wg3_launch(variant, m3, lds, (hipStream_t)stream);
      if (r != HDRSKY_OK) return r;
    }
    for (int q = 0; q < cnt; ++q) { const int rc = red.add(rjobs[q], rvec[q], S); if (rc != HDRSKY_OK) return rc; }
    for (int q = 0; q < cnt; ++q) done[members[base + q]] = true;
  }
  }
  return HDRSKY_OK;
}

}  // namespace

// Shared body of the two entry points.  ws == nullptr: split-K by fp32 atomics.  Otherwise deterministic: `ws` receives
// the per-workgroup partials (ws_floats floats available; when plan_only, nothing is launched and *need_floats returns
// the requirement), and a reduce launch per group adds them to dw / db in a fixed order.
static int wgrad_multi_impl(const hdrsky_wgrad_job* jobs, int njobs, float* ws, size_t ws_floats, bool plan_only,
                            size_t* need_floats, void* stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return HDRSKY_EINVAL;
  if (njobs > 256) return HDRSKY_EUNSUPPORTED;
  size_t ws_used = 0;
  // workgroups per launch: 128 for a layer on its own, 192 for a group.  Measured INSIDE the three-stream training step
  // (profiles/ab_bench.sh, HDRSKY_WGRAD=alone,0,group): a launch there does not have the chip to itself, and every pixel
  // chunk costs a partial slab that the reduce launch reads back - alone 256 (the best value for a launch timed on its own,
  // profiles/microbench_wgrad.py) -> 128 shortened the step by 3 %; 96 / 64 and group 160 / 224 / 256 were worse.
  const HdrskyHooks& hk = hdrsky_hooks();
  const int wg_hook = hk.wgrad_set ? hk.wgrad[0] : 0, force_small = hk.wgrad_set ? hk.wgrad[1] : 0,
            wg_hook_group = hk.wgrad_set ? hk.wgrad[2] : 0;                                   // (tuning hook HDRSKY_WGRAD)
  // wide stride-1 layers use 64x64-channel blocks when at least three of them share a launch
  Geo geo[256];
  bool done[256];
  for (int i = 0; i < njobs; ++i) done[i] = false;
  PendingReduce red;
  red.plan_only = plan_only; red.stream = stream;
  // layers with two final bf16 operands: the LDS-DMA ring kernel (deterministic mode only; HDRSKY_WGRAD2=0: A/B hook)
  const bool v2_on = hk.wgrad2 != 0;
  if (v2_on && (ws != nullptr || plan_only)) {
    const int rc2 = wgrad2_groups(jobs, njobs, done, ws, ws_floats, &ws_used, plan_only, stream, red);
    if (rc2 != HDRSKY_OK) return rc2;
  }
  // layers with a narrow side (<= 8 input or <= 4 output channels): their own kernel (HDRSKY_WGRAD3=0: A/B hook)
  const bool v3_on = hk.wgrad3 != 0;
  if (v3_on && (ws != nullptr || plan_only)) {
    const int rc3 = wgrad3_groups(jobs, njobs, done, ws, ws_floats, &ws_used, plan_only, stream, red);
    if (rc3 != HDRSKY_OK) return rc3;
  }
  { const int rcr = red.flush(); if (rcr != HDRSKY_OK) return rcr; }       // one reduce launch for both kinds of layer
  int nwide = 0;
  for (int i = 0; i < njobs; ++i) nwide += (!done[i] && can_go_big(jobs[i])) ? 1 : 0;
  const bool use_big = nwide >= 3 && !force_small;
  for (int i = 0; i < njobs; ++i) if (!done[i]) geo[i] = choose_geo(jobs[i], use_big && can_go_big(jobs[i]));
  for (int i = 0; i < njobs; ++i) {
    if (done[i]) continue;
    int members[256], nm = 0;
    double work[256], wsum = 0.0;
    for (int k = i; k < njobs; ++k)
      if (!done[k] && same_geo(geo[i], geo[k])) {
        const hdrsky_conv_desc& d = jobs[k].desc;
        work[nm] = (double)d.B * d.Ho * d.Wo * d.KH * d.KW * d.Cin * d.Cout;
        wsum += work[nm];
        members[nm++] = k;
        done[k] = true;
      }
    const int wg_total = nm == 1 ? (wg_hook > 0 ? wg_hook : 128) : (wg_hook_group > 0 ? wg_hook_group : (wg_hook > 0 ? wg_hook : 192));
    for (int base = 0; base < nm; base += WG_MAXJ) {   // kernel argument block holds WG_MAXJ jobs
      const int cnt = nm - base < WG_MAXJ ? nm - base : WG_MAXJ;
      MultiArgs m{};
      m.njobs = cnt;
      int lds = 0, rc = HDRSKY_OK;
      rc = with_variant(geo[i], [&](auto v) {
        using V = decltype(v);
        int blocks = 0, rblocks = 0, maxchunks = 1;
        for (int q = 0; q < cnt; ++q) {   // pass 1: geometry (the reduce launch's slice count follows the deepest split)
          WgradArgs tmp;
          if (fill_job(tmp, jobs[members[base + q]]) != HDRSKY_OK) continue;
          if (V::prepare(tmp, (int)(wg_total * work[base + q] / wsum + 0.5)) < 0) continue;
          if (tmp.nchunks > maxchunks) maxchunks = tmp.nchunks;
        }
        const int S = maxchunks >= 32 ? 16 : 4;
        for (int q = 0; q < cnt; ++q) {
          WgradArgs& a = m.job[q];
          int r = fill_job(a, jobs[members[base + q]]);
          if (r != HDRSKY_OK) return r;
          int target = (int)(wg_total * work[base + q] / wsum + 0.5);
          r = V::prepare(a, target);
          if (r < 0) return r;
          if (r > lds) lds = r;
          m.first[q] = blocks;
          blocks += a.cblocks * a.oblocks * a.nchunks;
          if (ws != nullptr || plan_only) {
            const size_t nslab = (size_t)a.nchunks * a.cblocks * a.oblocks * a.ntaps * V::CB * V::OB;
            const size_t nbias = (size_t)a.nchunks * a.oblocks * V::OB;
            a.ws = ws + ws_used;
            a.ws_db = a.db != nullptr ? ws + ws_used + nslab : nullptr;
            ws_used += nslab + (a.db != nullptr ? nbias : 0);
            m.rfirst[q] = rblocks;
            rblocks += (int)(((size_t)a.cblocks * a.oblocks * a.ntaps * V::CB * (V::OB / 4) + 256 / S - 1) / (256 / S));
          }
        }
        m.first[cnt] = blocks;
        m.rfirst[cnt] = rblocks;
        if (plan_only) { V::note(); if (ws != nullptr || plan_only) note_kernel("wgrad_reduce_kernel"); return (int)HDRSKY_OK; }
        if (ws != nullptr && ws_used > ws_floats) return (int)HDRSKY_EINVAL;
        int r = V::launch(m, lds, (hipStream_t)stream);
        if (r != HDRSKY_OK || ws == nullptr) return r;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, (hipStream_t)stream, m, V::CB, V::OB, S);
        HDRSKY_CHECK_LAUNCH();
        return (int)HDRSKY_OK;
      });
      if (rc != HDRSKY_OK) return rc;
    }
  }
  if (need_floats) *need_floats = ws_used;
  return HDRSKY_OK;
}

extern "C" int hdrsky_conv2d_wgrad_multi(const hdrsky_wgrad_job* jobs, int njobs, void* stream) {
  return wgrad_multi_impl(jobs, njobs, nullptr, 0, false, nullptr, stream);
}

extern "C" size_t hdrsky_conv2d_wgrad_ws_bytes(const hdrsky_wgrad_job* jobs, int njobs) {
  size_t need = 0;
  if (wgrad_multi_impl(jobs, njobs, nullptr, 0, true, &need, nullptr) != HDRSKY_OK) return 0;
  return (need > 0 ? need : 4) * sizeof(float);         // (0 = error: a call whose layers need no scratch still gets a token buffer)
}

// [host] the kernels one hdrsky_conv2d_wgrad_multi_det call on these jobs launches, " + "-separated, as rocprofv3 names them
extern "C" int hdrsky_conv2d_wgrad_kernel_names(const hdrsky_wgrad_job* jobs, int njobs, char* buf, int n) {
  if (!jobs || !buf || n <= 0) return HDRSKY_EINVAL;
  std::string names;
  g_wg_names = &names;
  size_t need = 0;
  const int rc = wgrad_multi_impl(jobs, njobs, nullptr, 0, true, &need, nullptr);
  g_wg_names = nullptr;
  snprintf(buf, (size_t)n, "%s", names.c_str());
  return rc;
}

// [host] 1 when the LDS-DMA kernel (conv_wgrad2_kernel) takes this job; with_final_bf16_x != 0: ... WOULD take it once x is
// replaced by the final bf16 tensor hdrsky_act_bf16 writes (the caller's question in front of that launch)
extern "C" int hdrsky_wgrad2_eligible(const hdrsky_wgrad_job* job, int with_final_bf16_x) {
  if (!job) return 0;
  if (!hdrsky_hooks().wgrad2) return 0;
  hdrsky_wgrad_job j = *job;
  if (with_final_bf16_x) { j.x_bf16 = 1; j.desc.in_mode = HDRSKY_IN_NONE; j.desc.in_slope = 1.f; }
  return v2_eligible(j) ? 1 : 0;
}

extern "C" int hdrsky_conv2d_wgrad_multi_det(const hdrsky_wgrad_job* jobs, int njobs, void* ws, size_t ws_bytes, void* stream) {
  if (!ws) return HDRSKY_EINVAL;
  return wgrad_multi_impl(jobs, njobs, (float*)ws, ws_bytes / sizeof(float), false, nullptr, stream);
}

extern "C" int hdrsky_conv2d_wgrad(const hdrsky_conv_desc* d, const float* x, const float* dy, const float* in_scale,
                                   const float* in_shift, const float* in_part, const float* in_gamma,
                                   const float* in_beta, float* dw, float* db, void* stream) {
  if (!d) return HDRSKY_EINVAL;
  hdrsky_wgrad_job j{};
  j.desc = *d;
  j.x = x; j.dy = dy; j.in_scale = in_scale; j.in_shift = in_shift; j.in_part = in_part; j.in_gamma = in_gamma;
  j.in_beta = in_beta; j.dw = dw; j.db = db;
  return hdrsky_conv2d_wgrad_multi(&j, 1, stream);
}
