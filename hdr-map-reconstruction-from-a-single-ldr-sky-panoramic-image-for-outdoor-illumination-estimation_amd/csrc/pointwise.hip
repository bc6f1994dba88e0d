// HBM/L2-bound kernels around the convolutions: InstanceNorm finalise / apply (+activation,
// residual, 2x2 max-pool), their backward (Grad-CAM sweep and training), soft-max and its
// picked-probability backward, Grad-CAM maps, the sun-radiance Dirac-delta head, tone mapping
// and alpha blending.  All fp32; float4 (16 B / lane) accesses; wave64 shuffles for reductions.
#include <cstdlib>

#include "common.h"
#include "hooks.h"

namespace {

// InstanceNorm scale/shift for sample b from the producer conv's per-tile (sum, sumsq)
// partials [B][nparts][2][C] - identical summation order to conv_igemm's prologue.
__device__ __forceinline__ void in_tables_from_partials(const float* __restrict__ part, int b, int nparts, int C,
                                                        float inv_count, const float* gamma, const float* beta,
                                                        float eps, float* sScale, float* sShift, float* sMean,
                                                        float* sInv) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s, ss;
    in_partial_sums(part + (size_t)b * nparts * 2 * C + c, nparts, C, s, ss);
    const float mean = s * inv_count;
    const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
    const float rstd = 1.f / sqrtf(var + eps);
    const float inv = gamma[c] * rstd;
    sScale[c] = inv;
    sShift[c] = beta[c] - mean * inv;
    if (sMean) { sMean[c] = mean; sInv[c] = rstd; }
  }
}

// ---------------------------------------------------------------------------------------------
// y = leaky(IN(x)) [+ residual]; optional 2x2/2 max-pool of y.   generator.py:26-35,98-106 ;
// sunpose_net.py:20-30,55-62 (ops.maxpool2d ops.py:299-300)
// grid = B * S blocks; block handles a contiguous range of 2x2 windows (pool) or pixels.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                         int nparts, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float slope,
                                                         const float* __restrict__ residual, float* __restrict__ y,
                                                         float* __restrict__ ypool, int B, int H, int W, int C, int S, int x16) {
  extern __shared__ float sm[];
  float* sScale = sm;
  float* sShift = sm + C;
  const int b = blockIdx.x / S, s = blockIdx.x % S;
  in_tables_from_partials(part, b, nparts, C, 1.f / (float)(H * W), gamma, beta, eps, sScale, sShift, nullptr, nullptr);
  __syncthreads();
  const int c4 = C >> 2;
  const size_t xo = (size_t)b * H * W * C;         // (x: fp32 or - x16 - bf16 storage of the raw conv output)
  float* yb = y + (size_t)b * H * W * C;
  const float* rb = residual ? residual + (size_t)b * H * W * C : nullptr;
  if (ypool == nullptr) {
    const int total = H * W * c4;
    const int per = (total + S - 1) / S;
    const int end = min(total, (s + 1) * per);
    for (int i = s * per + threadIdx.x; i < end; i += 256) {
      const int c = (i % c4) * 4;
      float4 v = ld4any(x, x16, xo + (size_t)i * 4);
      v.x = leaky(v.x * sScale[c] + sShift[c], slope);
      v.y = leaky(v.y * sScale[c + 1] + sShift[c + 1], slope);
      v.z = leaky(v.z * sScale[c + 2] + sShift[c + 2], slope);
      v.w = leaky(v.w * sScale[c + 3] + sShift[c + 3], slope);
      if (rb) {
        const float4 r = reinterpret_cast<const float4*>(rb)[i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      reinterpret_cast<float4*>(yb)[i] = v;
    }
  } else {
    const int Hp = H >> 1, Wp = W >> 1;
    float* pb = ypool + (size_t)b * Hp * Wp * C;
    const int total = Hp * Wp * c4;
    const int per = (total + S - 1) / S;
    const int end = min(total, (s + 1) * per);
    for (int i = s * per + threadIdx.x; i < end; i += 256) {
      const int cq = i % c4, pw = (i / c4) % Wp, ph = i / (c4 * Wp);
      const int c = cq * 4;
      float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const size_t idx = ((size_t)(2 * ph + dy) * W + (2 * pw + dx)) * c4 + cq;
          float4 v = ld4any(x, x16, xo + idx * 4);
          v.x = leaky(v.x * sScale[c] + sShift[c], slope);
          v.y = leaky(v.y * sScale[c + 1] + sShift[c + 1], slope);
          v.z = leaky(v.z * sScale[c + 2] + sShift[c + 2], slope);
          v.w = leaky(v.w * sScale[c + 3] + sShift[c + 3], slope);
          if (rb) {
            const float4 r = reinterpret_cast<const float4*>(rb)[idx];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          reinterpret_cast<float4*>(yb)[idx] = v;
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      reinterpret_cast<float4*>(pb)[i] = m;
    }
  }
}

// mean / rstd / scale / shift tables [B][C] (needed by backward kernels and host-side checks)
// scale / shift [B][C] of the fused operand transform, in the formula of the conv prologues (gamma / sqrt(var + eps)):
// what every workgroup of a consumer launch computes for itself in HDRSKY_IN_PARTIALS mode, once per tensor instead - a
// 128x512 map has 512 tile partials per sample and a 64-channel layer on it 4096 workgroups (1 GB of L2 reads for tables)
__global__ void __launch_bounds__(256) in_affine_kernel(const float* __restrict__ part, int nparts, int B, int C, float inv_count,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                        const float* __restrict__ gamma2 = nullptr, const float* __restrict__ beta2 = nullptr,
                                                        int gsplit = 0) {
  // block = (sample, 32 channels) x 8 slices of the tile range: slice q adds the tiles [q * per, (q + 1) * per) in order
  // (loads eight at a time), then the slices are added in slice order - a fixed order
  __shared__ float sS[8][32], sQ[8][32];
  const int cb = C / 32 + ((C % 32) ? 1 : 0);
  const int b = blockIdx.x / cb, c = (blockIdx.x % cb) * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
  const int per = (nparts + 7) / 8, p0 = q * per, np = max(0, min(per, nparts - p0));
  float s = 0.f, ss = 0.f;
  if (c < C && np > 0) in_partial_sums(part + ((size_t)b * nparts + p0) * 2 * C + c, np, C, s, ss);
  sS[q][threadIdx.x & 31] = s; sQ[q][threadIdx.x & 31] = ss;
  __syncthreads();
  if (q != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) { s += sS[k][threadIdx.x]; ss += sQ[k][threadIdx.x]; }
  const float mean = s * inv_count;
  const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
  const bool g1 = gsplit > 0 && b >= gsplit;      // paired tensors: the second half of the batch is another layer's (its own gamma / beta)
  const float inv = (g1 ? gamma2 : gamma)[c] / sqrtf(var + eps);
  scale[(size_t)b * C + c] = inv;
  shift[(size_t)b * C + c] = (g1 ? beta2 : beta)[c] - mean * inv;
}

__global__ void in_finalize_kernel(const float* __restrict__ part, int nparts, int B, int C, float inv_count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float* mean, float* rstd, float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  float s, ss;
  in_partial_sums(part + (size_t)b * nparts * 2 * C + c, nparts, C, s, ss);
  const float m = s * inv_count;
  const float var = fmaxf(ss * inv_count - m * m, 0.f);
  const float r = 1.f / sqrtf(var + eps);
  if (mean) mean[i] = m;
  if (rstd) rstd[i] = r;
  const float inv = gamma[c] * r;
  if (scale) scale[i] = inv;
  if (shift) shift[i] = beta[c] - m * inv;
}

// Keras BatchNormalization inference-mode affine: scale = gamma*rsqrt(mv+eps), shift = beta - mm*scale
__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* mm, const float* mv,
                                      float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float inv = gamma[c] / sqrtf(mv[c] + eps);
  scale[c] = inv;
  shift[c] = beta[c] - mm[c] * inv;
}

// ---------------------------------------------------------------------------------------------
// Backward of y = leaky(IN(x)) (optionally followed by the 2x2 max-pool), data gradient only:
//   g  = upstream routed through pool-argmax (first max in scan order) and the activation mask
//   dx = gamma*rstd * (g - mean_hw(g) - xhat*mean_hw(g*xhat))
// A block owns (sample, 16-channel group, spatial slice).  MODE 0 writes the slice's (sum g, sum g*xhat) to the
// workspace, MODE 1 adds the slices of its sample in a fixed order and applies the formula to its slice, MODE 2
// (one slice) does both in one launch.  The per-(b,c) sums are also the (dbeta, dgamma) contributions.
// Used by the Grad-CAM sweep (grad_cam.py:31 tf.gradients through sunpose_net.py:20-30,55-62) and the IN backward
// of the training step (train.py:402).
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) norm_act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                           int nparts, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float slope,
                                                           const float* __restrict__ dy, int pooled,
                                                           void* __restrict__ dxv, int dx_bf16, float* __restrict__ sums,
                                                           float* dgamma, float* dbeta, float* __restrict__ ws, int S,
                                                           int B, int H, int W, int C, const float* __restrict__ gamma2 = nullptr,
                                                           const float* __restrict__ beta2 = nullptr, int gsplit = 0) {
  __shared__ float sRed[64][2][16];
  __shared__ float sM[2][16];
  const int groups = C >> 4;
  const int sl_id = blockIdx.x % S;
  const int b = (blockIdx.x / S) / groups, cg = (blockIdx.x / S) % groups;
  const int cl = (threadIdx.x & 3) * 4;      // channel offset inside the group (float4)
  const int slot = threadIdx.x >> 2;         // 64 pixel slots
  const int c = cg * 16 + cl;
  // per-channel constants (4 channels per thread).  The 16 channels' statistics are reduced cooperatively: thread
  // (slice, ch) sums every 16th partial, then 16 threads add the 16 slices in a fixed order (one memory round trip
  // instead of nparts dependent ones).
  float mean[4], rstd[4], gm[4], bt[4];
  {
    __shared__ float sPart[16][16][2];
    __shared__ float sMR[16][2];
    const int ch = threadIdx.x & 15, sl = threadIdx.x >> 4;
    float s = 0.f, ss = 0.f;
    const float* pp = part + (size_t)b * nparts * 2 * C + cg * 16 + ch;
    for (int p = sl; p < nparts; p += 16) { s += pp[(2 * p) * C]; ss += pp[(2 * p + 1) * C]; }
    sPart[sl][ch][0] = s; sPart[sl][ch][1] = ss;
    __syncthreads();
    if (threadIdx.x < 16) {
      float ts = 0.f, tss = 0.f;
      for (int k = 0; k < 16; ++k) { ts += sPart[k][threadIdx.x][0]; tss += sPart[k][threadIdx.x][1]; }
      const float inv_count = 1.f / (float)(H * W);
      const float m = ts * inv_count;
      const float var = fmaxf(tss * inv_count - m * m, 0.f);
      sMR[threadIdx.x][0] = m; sMR[threadIdx.x][1] = 1.f / sqrtf(var + eps);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mean[j] = sMR[cl + j][0]; rstd[j] = sMR[cl + j][1];
      const bool g1 = gsplit > 0 && b >= gsplit;      // paired tensors (hdrsky_norm_act_bwd_pair): the second half's own gamma / beta
      gm[j] = (g1 ? gamma2 : gamma)[c + j]; bt[j] = (g1 ? beta2 : beta)[c + j];
    }
  }
  const size_t xo = (size_t)b * H * W * C + c;
  // dy: fp32, or bf16 (bit 1 of the flag word) when it is the output of a data-gradient conv that nothing else reads;
  // x: fp32, or bf16 (bit 2) - the raw conv output in front of the norm layer as the single-product mode stores it
  const bool dy16 = (dx_bf16 & 2) != 0;
  const int x16 = dx_bf16 & 4;
  dx_bf16 &= 1;
  // dx: fp32, or bf16 when its only readers are a data-gradient conv and a weight gradient (both round it to bf16 anyway)
  float* dxb = reinterpret_cast<float*>(dxv) + (size_t)b * H * W * C + c;
  unsigned short* dxh = reinterpret_cast<unsigned short*>(dxv) + (size_t)b * H * W * C + c;
  auto put = [&](size_t off, const float4& o) {
    if (dx_bf16)
      *reinterpret_cast<uint2*>(dxh + off) = uint2{(unsigned)f2bf(o.x) | ((unsigned)f2bf(o.y) << 16), (unsigned)f2bf(o.z) | ((unsigned)f2bf(o.w) << 16)};
    else
      *reinterpret_cast<float4*>(dxb + off) = o;
  };
  const int Hp = H >> 1, Wp = W >> 1;
  const int nunits = pooled ? Hp * Wp : H * W;            // pixels, or 2x2 windows
  const int per = (nunits + S - 1) / S;
  const int u0 = sl_id * per, u1 = min(nunits, u0 + per);
  const float* dyb = dy + (size_t)b * nunits * C + c;
  const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy) + (size_t)b * nunits * C + c;
  auto get_dy = [&](size_t off) {
    if (dy16) {
      const uint2 u = *reinterpret_cast<const uint2*>(dyh + off);
      return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                         __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(dyb + off);
  };

  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
  for (int pass = (MODE == 1 ? 1 : 0); pass < (MODE == 0 ? 1 : 2); ++pass) {
    if (pass == 1 && MODE == 1) {
      // totals of this (sample, channel group) over the S slices, fixed order
      if (threadIdx.x < 32) {
        const int which = threadIdx.x >> 4, ch = threadIdx.x & 15;
        float t = 0.f;
        for (int k = 0; k < S; ++k) t += ws[(((size_t)b * S + k) * 2 + which) * C + cg * 16 + ch];
        sM[which][ch] = t;
        if (sl_id == 0) {
          if (sums) sums[((size_t)b * 2 + which) * C + cg * 16 + ch] = t;
          if (which == 0 && dbeta) atomicAdd(dbeta + cg * 16 + ch, t);
          if (which == 1 && dgamma) atomicAdd(dgamma + cg * 16 + ch, t);
        }
      }
      __syncthreads();
      const float inv_count = 1.f / (float)(H * W);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] = sM[0][cl + j] * inv_count; s2[j] = sM[1][cl + j] * inv_count; }
    }
    if (!pooled) {
#pragma unroll 4
      for (int p = u0 + slot; p < u1; p += 64) {
        const float4 xv = ld4any(x, x16, xo + (size_t)p * C);
        const float4 up = get_dy((size_t)p * C);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float us[4] = {up.x, up.y, up.z, up.w};
        float xh[4], g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[j] = (xs[j] - mean[j]) * rstd[j];
          const float pre = xh[j] * gm[j] + bt[j];
          g[j] = us[j] * (pre > 0.f ? 1.f : slope);
        }
        if (pass == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { s1[j] += g[j]; s2[j] += g[j] * xh[j]; }
        } else {
          float4 o;
          o.x = gm[0] * rstd[0] * (g[0] - s1[0] - xh[0] * s2[0]);
          o.y = gm[1] * rstd[1] * (g[1] - s1[1] - xh[1] * s2[1]);
          o.z = gm[2] * rstd[2] * (g[2] - s1[2] - xh[2] * s2[2]);
          o.w = gm[3] * rstd[3] * (g[3] - s1[3] - xh[3] * s2[3]);
          put((size_t)p * C, o);
        }
      }
    } else {
#pragma unroll 2
      for (int pw = u0 + slot; pw < u1; pw += 64) {
        const int ph = pw / Wp, px = pw % Wp;
        const float4 up = get_dy((size_t)pw * C);
        const float us[4] = {up.x, up.y, up.z, up.w};
        float4 xv[4];
        float act[4][4], xh[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = (2 * ph + (k >> 1)) * W + 2 * px + (k & 1);
          xv[k] = ld4any(x, x16, xo + (size_t)p * C);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xs[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            xh[k][j] = (xs[j] - mean[j]) * rstd[j];
            act[k][j] = leaky(xh[k][j] * gm[j] + bt[j], slope);
          }
        }
        float g[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int am = 0;
          float best = act[0][j];
#pragma unroll
          for (int k = 1; k < 4; ++k)
            if (act[k][j] > best) { best = act[k][j]; am = k; }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float pre = xh[k][j] * gm[j] + bt[j];
            g[k][j] = (k == am) ? us[j] * (pre > 0.f ? 1.f : slope) : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (pass == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += g[k][j]; s2[j] += g[k][j] * xh[k][j]; }
          } else {
            const int p = (2 * ph + (k >> 1)) * W + 2 * px + (k & 1);
            float4 o;
            o.x = gm[0] * rstd[0] * (g[k][0] - s1[0] - xh[k][0] * s2[0]);
            o.y = gm[1] * rstd[1] * (g[k][1] - s1[1] - xh[k][1] * s2[1]);
            o.z = gm[2] * rstd[2] * (g[k][2] - s1[2] - xh[k][2] * s2[2]);
            o.w = gm[3] * rstd[3] * (g[k][3] - s1[3] - xh[k][3] * s2[3]);
            put((size_t)p * C, o);
          }
        }
      }
    }
    if (pass == 0) {
      // block reduction over the 64 pixel slots (fixed order -> deterministic)
#pragma unroll
      for (int j = 0; j < 4; ++j) { sRed[slot][0][cl + j] = s1[j]; sRed[slot][1][cl + j] = s2[j]; }
      __syncthreads();
      if (threadIdx.x < 32) {
        const int which = threadIdx.x >> 4, ch = threadIdx.x & 15;
        float t = 0.f;
        for (int k = 0; k < 64; ++k) t += sRed[k][which][ch];
        if (MODE == 0) {
          ws[(((size_t)b * S + sl_id) * 2 + which) * C + cg * 16 + ch] = t;
        } else {
          sM[which][ch] = t;
          if (sums) sums[((size_t)b * 2 + which) * C + cg * 16 + ch] = t;
          if (which == 0 && dbeta) atomicAdd(dbeta + cg * 16 + ch, t);
          if (which == 1 && dgamma) atomicAdd(dgamma + cg * 16 + ch, t);
        }
      }
      if (MODE == 2) {
        __syncthreads();
        const float inv_count = 1.f / (float)(H * W);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[j] = sM[0][cl + j] * inv_count; s2[j] = sM[1][cl + j] * inv_count; }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same backward in ONE launch and ONE read of (x, dy) for maps whose (sample, 16-channel group) slab fits the registers
// of a workgroup (round 5; the sliced form above reads x and dy twice and is two launches from 512 pixels per slice on: 20 of
// the training step's 29 norm_act_bwd launches).  Block = (sample, 16-channel group), NT threads; a thread holds NV units
// (pixels, or 2x2 windows of the pooled form) of one float4 channel column as xhat in registers with the upstream gradient
// beside it, the two sums are reduced over the block in a fixed order (xor shuffles inside a wave, waves in index order),
// and the formula is applied to the registers.  Same arithmetic per element as norm_act_bwd_kernel; the sums are added in
// another (still fixed) order, so results differ from the sliced form by fp32 rounding of those two sums only.
// ---------------------------------------------------------------------------------------------
template <int NT, int NV, int CG, bool POOLED, bool DY16>
__global__ void __launch_bounds__(NT) norm_act_bwd1_kernel(const float* __restrict__ x, const float* __restrict__ part, int nparts,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float slope, const float* __restrict__ dy, void* __restrict__ dxv,
                                                           int flags, float* __restrict__ sums, float* dgamma, float* dbeta, int H,
                                                           int W, int C, const float* __restrict__ gamma2 = nullptr,
                                                           const float* __restrict__ beta2 = nullptr, int gsplit = 0) {
  constexpr int NW = NT / 64, NC = CG / 4, NS = NT / NC;      // waves; float4 columns per pixel; pixel (window) slots per pass
  static_assert(NT >= 16 * CG && (CG == 8 || CG == 16), "statistics prologue: 16 partial slices x CG channels");
  __shared__ float sPart[16][CG][2];
  __shared__ float sMR[CG][2];
  __shared__ float sRed[NW][2][CG];
  __shared__ float sM[2][CG];
  const int groups = C / CG;
  const int b = blockIdx.x / groups, cg = blockIdx.x % groups;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = (tid % NC) * 4, slot = tid / NC;
  const int c = cg * CG + cl;
  const bool dx_bf16 = (flags & 1) != 0;
  const int x16 = flags & 4;
  // statistics of the CG channels from the forward conv's tile partials (as in norm_act_bwd_kernel)
  if (tid < 16 * CG) {
    const int ch = tid % CG, sl = tid / CG;
    float s = 0.f, ss = 0.f;
    const float* pp = part + (size_t)b * nparts * 2 * C + cg * CG + ch;
    for (int p = sl; p < nparts; p += 16) { s += pp[(2 * p) * C]; ss += pp[(2 * p + 1) * C]; }
    sPart[sl][ch][0] = s; sPart[sl][ch][1] = ss;
  }
  __syncthreads();
  if (tid < CG) {
    float ts = 0.f, tss = 0.f;
    for (int k = 0; k < 16; ++k) { ts += sPart[k][tid][0]; tss += sPart[k][tid][1]; }
    const float inv_count = 1.f / (float)(H * W);
    const float m = ts * inv_count;
    const float var = fmaxf(tss * inv_count - m * m, 0.f);
    sMR[tid][0] = m; sMR[tid][1] = 1.f / sqrtf(var + eps);
  }
  __syncthreads();
  float mean[4], rstd[4], gm[4], bt[4];
  const bool g1 = gsplit > 0 && b >= gsplit;      // paired tensors (hdrsky_norm_act_bwd_pair): the second half's own gamma / beta
#pragma unroll
  for (int j = 0; j < 4; ++j) { mean[j] = sMR[cl + j][0]; rstd[j] = sMR[cl + j][1]; gm[j] = (g1 ? gamma2 : gamma)[c + j]; bt[j] = (g1 ? beta2 : beta)[c + j]; }
  const int Wp = W >> 1;
  const int nunits = POOLED ? (H >> 1) * Wp : H * W;
  // workgroup-uniform bases + 32-bit element offsets (a sample's slab is < 2^31 elements): one address register per load
  const float* xs32 = x + (size_t)b * H * W * C;
  const unsigned short* xs16 = reinterpret_cast<const unsigned short*>(x) + (size_t)b * H * W * C;
  const float* dyb = dy + (size_t)b * nunits * C;
  const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy) + (size_t)b * nunits * C;
  constexpr int PX = POOLED ? 4 : 1;             // pixels per unit
  auto pix_off = [&](int u, int q) -> unsigned {
    if (!POOLED) return (unsigned)u * (unsigned)C + (unsigned)c;
    const int ph = u / Wp, px = u - ph * Wp;
    return (unsigned)((2 * ph + (q >> 1)) * W + 2 * px + (q & 1)) * (unsigned)C + (unsigned)c;
  };
  float xh[NV][PX][4];
  uint2 u16[DY16 ? NV : 1];
  float4 u32[DY16 ? 1 : NV];
  // ---- one read of x and dy: every load of the thread issued before the first use --------------------------------
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const unsigned uo = (unsigned)(slot + k * NS) * (unsigned)C + (unsigned)c;
    if (DY16) u16[k] = *reinterpret_cast<const uint2*>(dyh + uo);
    else u32[k] = *reinterpret_cast<const float4*>(dyb + uo);
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const unsigned o = pix_off(slot + k * NS, q);
      float4 v;
      if (x16) {
        const uint2 t = *reinterpret_cast<const uint2*>(xs16 + o);
        v = make_float4(__builtin_bit_cast(float, t.x << 16), __builtin_bit_cast(float, t.x & 0xffff0000u),
                        __builtin_bit_cast(float, t.y << 16), __builtin_bit_cast(float, t.y & 0xffff0000u));
      } else if (flags & 8) {      // (tuning hook HDRSKY_NAB_NT: x - a raw conv output of the forward pass, read here for the last time)
        typedef __attribute__((ext_vector_type(4))) float f32v4_t;
        const f32v4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32v4_t*>(xs32 + o));
        v = make_float4(t.x, t.y, t.z, t.w);
      } else {
        v = *reinterpret_cast<const float4*>(xs32 + o);
      }
      xh[k][q][0] = v.x; xh[k][q][1] = v.y; xh[k][q][2] = v.z; xh[k][q][3] = v.w;
    }
  }
  auto up_of = [&](int k, float (&us)[4]) {
    if (DY16) {
      us[0] = __builtin_bit_cast(float, u16[k].x << 16); us[1] = __builtin_bit_cast(float, u16[k].x & 0xffff0000u);
      us[2] = __builtin_bit_cast(float, u16[k].y << 16); us[3] = __builtin_bit_cast(float, u16[k].y & 0xffff0000u);
    } else {
      us[0] = u32[k].x; us[1] = u32[k].y; us[2] = u32[k].z; us[3] = u32[k].w;
    }
  };
  // g of unit k, pixel q, channel j (pool routing to the first maximum in scan order, activation mask)
  auto grads = [&](int k, float (&g)[PX][4]) {
    float us[4];
    up_of(k, us);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (POOLED) {
        int am = 0;
        float best = leaky(xh[k][0][j] * gm[j] + bt[j], slope);
#pragma unroll
        for (int q = 1; q < PX; ++q) {
          const float a = leaky(xh[k][q][j] * gm[j] + bt[j], slope);
          if (a > best) { best = a; am = q; }
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) g[q][j] = (q == am) ? us[j] * ((xh[k][q][j] * gm[j] + bt[j]) > 0.f ? 1.f : slope) : 0.f;
      } else {
        g[0][j] = us[j] * ((xh[k][0][j] * gm[j] + bt[j]) > 0.f ? 1.f : slope);
      }
    }
  };
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int q = 0; q < PX; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) xh[k][q][j] = (xh[k][q][j] - mean[j]) * rstd[j];
    float g[PX][4];
    grads(k, g);
#pragma unroll
    for (int q = 0; q < PX; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] += g[q][j]; s2[j] += g[q][j] * xh[k][q][j]; }
  }
  // ---- block sums, fixed order: lanes of equal channel column inside the wave, then the waves in index order ----------
#pragma unroll
  for (int o = NC; o < 64; o <<= 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
  if (lane < NC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { sRed[wave][0][cl + j] = s1[j]; sRed[wave][1][cl + j] = s2[j]; }
  }
  __syncthreads();
  if (tid < 2 * CG) {
    const int which = tid / CG, ch = tid % CG;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += sRed[w][which][ch];
    sM[which][ch] = t;
    if (sums) sums[((size_t)b * 2 + which) * C + cg * CG + ch] = t;
    if (which == 0 && dbeta) atomicAdd(dbeta + cg * CG + ch, t);
    if (which == 1 && dgamma) atomicAdd(dgamma + cg * CG + ch, t);
  }
  __syncthreads();
  const float inv_count = 1.f / (float)(H * W);
#pragma unroll
  for (int j = 0; j < 4; ++j) { s1[j] = sM[0][cl + j] * inv_count; s2[j] = sM[1][cl + j] * inv_count; }
  // ---- apply from the registers ----------------------------------------------------------------------------------
  float* dxb = reinterpret_cast<float*>(dxv) + (size_t)b * H * W * C;
  unsigned short* dxh = reinterpret_cast<unsigned short*>(dxv) + (size_t)b * H * W * C;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float g[PX][4];
    grads(k, g);
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const unsigned po = pix_off(slot + k * NS, q);
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = gm[j] * rstd[j] * (g[q][j] - s1[j] - xh[k][q][j] * s2[j]);
      if (dx_bf16)
        *reinterpret_cast<uint2*>(dxh + po) =
            uint2{(unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16), (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16)};
      else
        *reinterpret_cast<float4*>(dxb + po) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// soft-max head of the sun-pose net (sunpose_net.py:64-70): z = relu(sum_s part[s] + bias),
// cmf = softmax(z); also the running global max of cmf (generator.py:160 reduce_max over the
// whole batch tensor) via an order-independent integer atomicMax on the (positive) float bits.
// One block per row.
// ---------------------------------------------------------------------------------------------
// One 1024-thread block per row; a thread owns float4 groups n4 = tid, tid + 1024, ... (RV of them, 4 at N = 16384:
// everything stays in registers, the split-R partials of a group are 16-byte loads issued together).
// PICK: also the Grad-CAM seed d y_c / d z of softmax_pick_bwd_kernel below (same arithmetic, same first-arg-max rule), from
// the row that is in registers anyway - one launch less on the forward pass's critical chain.  pick_src == nullptr: the
// row's own cmf picks the class (inference.py:98), else pick_src[m, :] (sunpose_gt in training, train.py:265-267).
template <int RV, bool PICK>
__global__ void __launch_bounds__(1024) softmax_head_kernel(const float* __restrict__ part, int nsplit, int M, int N,
                                                            const float* __restrict__ bias, float* __restrict__ z,
                                                            float* __restrict__ cmf, unsigned int* gmax_bits,
                                                            const float* __restrict__ pick_src, float* __restrict__ dz,
                                                            int* __restrict__ idx_out) {
  __shared__ float sred[16];
  __shared__ int sidx[16];
  __shared__ float syc;
  const int m = blockIdx.x, tid = threadIdx.x, n4 = N >> 2;
  float4 v[RV];
  unsigned zpos = 0u;                                 // PICK: bit 4r + j = [z > 0] of this thread's element (r, j)
  float lmax = 0.f;                                   // relu output: >= 0
#pragma unroll
  for (int r = 0; r < RV; ++r) {
    const int q = tid + r * 1024;
    v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < n4) {
      float4 a = bias ? reinterpret_cast<const float4*>(bias)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < nsplit; ++s) {              // summation order as in fc_finalize: bias, then slices ascending
        const float4 t = reinterpret_cast<const float4*>(part + ((size_t)s * M + m) * N)[q];
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
      v[r] = a;
      if (PICK) zpos |= ((a.x > 0.f ? 1u : 0u) | (a.y > 0.f ? 2u : 0u) | (a.z > 0.f ? 4u : 0u) | (a.w > 0.f ? 8u : 0u)) << (4 * r);
      if (z) reinterpret_cast<float4*>(z + (size_t)m * N)[q] = a;
      lmax = fmaxf(lmax, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
    }
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) sred[tid >> 6] = lmax;
  __syncthreads();
  float rmax = sred[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) rmax = fmaxf(rmax, sred[w]);
  __syncthreads();
  float lsum = 0.f;
#pragma unroll
  for (int r = 0; r < RV; ++r) {
    if (tid + r * 1024 < n4) {
      v[r].x = expf(v[r].x - rmax); v[r].y = expf(v[r].y - rmax); v[r].z = expf(v[r].z - rmax); v[r].w = expf(v[r].w - rmax);
      lsum += (v[r].x + v[r].y) + (v[r].z + v[r].w);
    }
  }
  lsum = wave_sum(lsum);
  if ((tid & 63) == 0) sred[tid >> 6] = lsum;
  __syncthreads();
  float rsum = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) rsum += sred[w];
  float pmax = 0.f;
#pragma unroll
  for (int r = 0; r < RV; ++r) {
    const int q = tid + r * 1024;
    if (q < n4) {
      const float4 pv = make_float4(v[r].x / rsum, v[r].y / rsum, v[r].z / rsum, v[r].w / rsum);
      reinterpret_cast<float4*>(cmf + (size_t)m * N)[q] = pv;
      pmax = fmaxf(pmax, fmaxf(fmaxf(pv.x, pv.y), fmaxf(pv.z, pv.w)));
      if (PICK) v[r] = pv;
    }
  }
  if (gmax_bits) {
    pmax = wave_max(pmax);
    if ((tid & 63) == 0) atomicMax(gmax_bits, __float_as_uint(pmax));
  }
  if (PICK) {
    // first arg-max of the picking row: ascending element index per thread with a strict >, ties between threads to the
    // smaller index (softmax_pick_bwd_kernel's rule)
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < RV; ++r) {
      const int q = tid + r * 1024;
      if (q < n4) {
        const float4 pk = pick_src ? reinterpret_cast<const float4*>(pick_src + (size_t)m * N)[q] : v[r];
        const float e[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (e[j] > best) { best = e[j]; bi = 4 * q + j; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    __syncthreads();                                  // (sred was read above by every thread)
    if ((tid & 63) == 0) { sred[tid >> 6] = best; sidx[tid >> 6] = bi; }
    __syncthreads();
    best = sred[0]; bi = sidx[0];
    for (int w = 1; w < 16; ++w)
      if (sred[w] > best || (sred[w] == best && sidx[w] < bi)) { best = sred[w]; bi = sidx[w]; }
#pragma unroll
    for (int r = 0; r < RV; ++r) {                    // the owner of element bi publishes y_c = cmf[m, bi]
      const int q = tid + r * 1024;
      if (q == (bi >> 2)) { const float e[4] = {v[r].x, v[r].y, v[r].z, v[r].w}; syc = e[bi & 3]; }
    }
    __syncthreads();
    const float yc = syc;
    if (tid == 0 && idx_out) idx_out[m] = bi;
#pragma unroll
    for (int r = 0; r < RV; ++r) {
      const int q = tid + r * 1024;
      if (q < n4) {
        const float p4[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
        float g4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = yc * ((4 * q + j == bi ? 1.f : 0.f) - p4[j]);
          g4[j] = ((zpos >> (4 * r + j)) & 1u) ? g : 0.f;
        }
        reinterpret_cast<float4*>(dz + (size_t)m * N)[q] = make_float4(g4[0], g4[1], g4[2], g4[3]);
      }
    }
  }
}

// d y_c / d z for y_c = cmf[m, idx_m] through softmax and the relu in front of it:
//   dz_j = y_c * ((j == idx) - cmf_j) * [z_j > 0]
// idx_m = first argmax of pick_src[m, :] (cmf itself at inference, inference.py:98;
// sunpose_gt in training, train.py:265-267).
__global__ void __launch_bounds__(1024) softmax_pick_bwd_kernel(const float* __restrict__ cmf,
                                                                const float* __restrict__ z,
                                                                const float* __restrict__ pick_src, int N,
                                                                float* __restrict__ dz, int* __restrict__ idx_out) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const int m = blockIdx.x, tid = threadIdx.x;
  const float* pr = pick_src + (size_t)m * N;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int n = tid; n < N; n += 1024) {           // ascending n per thread: strict > keeps the first maximum
    const float v = pr[n];
    if (v > best) { best = v; bi = n; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  best = sv[0]; bi = si[0];
  for (int w = 1; w < 16; ++w)
    if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
  const float yc = cmf[(size_t)m * N + bi];
  if (tid == 0 && idx_out) idx_out[m] = bi;
  for (int n = tid; n < N; n += 1024) {
    const float p = cmf[(size_t)m * N + n];
    const float g = yc * ((n == bi ? 1.f : 0.f) - p);
    dz[(size_t)m * N + n] = z[(size_t)m * N + n] > 0.f ? g : 0.f;
  }
}

// y[m][n] = (sum_s part[s][m][n] + bias[n]) with optional relu and optional mask [mask_src > 0]
__global__ void fc_finalize_kernel(const float* __restrict__ part, int nsplit, int M, int N,
                                   const float* __restrict__ bias, int relu, const float* __restrict__ mask_src,
                                   float* __restrict__ y, unsigned int* __restrict__ zero_word) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && zero_word) *zero_word = 0u;      // the max accumulator of a later hdrsky_softmax_head: no memset launch
  if (i >= M * N) return;
  float v = bias ? bias[i % N] : 0.f;
  for (int s = 0; s < nsplit; ++s) v += part[(size_t)s * M * N + i];
  if (relu) v = fmaxf(v, 0.f);
  if (mask_src) v = mask_src[i] > 0.f ? v : 0.f;
  y[i] = v;
}

// out[b][c] = scale * sum_p x[b][p][c]     (GAP of the Grad-CAM gradient, grad_cam.py:34)
__global__ void __launch_bounds__(256) spatial_sum_kernel(const float* __restrict__ x, int P, int C, float scale,
                                                          float* __restrict__ out) {
  __shared__ float sred[256];
  const int b = blockIdx.x;
  const int c = threadIdx.x % C;           // C in {32,64,128} divides 256
  const int lanes_p = 256 / C;
  const int p0 = threadIdx.x / C;
  float s = 0.f;
  for (int p = p0; p < P; p += lanes_p) s += x[((size_t)b * P + p) * C + c];
  sred[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < lanes_p; ++k) t += sred[k * C + threadIdx.x];
    out[(size_t)b * C + threadIdx.x] = t * scale;
  }
}

// cam[b][p] = relu(sum_c w[b][c] * A[b][p][c])   (grad_cam.py:35-38); one wave per pixel group
__device__ __forceinline__ void cam_body(const float* __restrict__ A, const float* __restrict__ w, int w_nparts,
                                         float w_scale, int P, int C, float* __restrict__ cam, float* sw, int b, int bx, int gdx) {
  // w_nparts == 0: w is a [B][C] table; > 0: w is a conv statistics tensor [B][nparts][2][C] whose sum plane is
  // reduced here (the GAP of the activation gradient, grad_cam.py:34); < 0: w is the gradient map itself,
  // [B][-nparts pixels][C] (a small one: every block of a sample repeats the sum)
  if (w_nparts < 0) {
    // the map's pixels are dealt to 256 / (C/4) slices of threads (16-byte loads, 4 channels per thread), then the slices
    // are added in slice order: one thread per channel walking all pixels was a 256-deep chain of dependent loads (18 us)
    const int np = -w_nparts, c4n = C >> 2, nsl = 256 / c4n, cq = threadIdx.x % c4n, sl = threadIdx.x / c4n;
    float* sred = sw + C;                       // [nsl][C]
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = sl; p < np; p += nsl) {
      const float4 v = *reinterpret_cast<const float4*>(w + ((size_t)b * np + p) * C + cq * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(sred + sl * C + cq * 4) = acc;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float t = 0.f;
      for (int k = 0; k < nsl; ++k) t += sred[k * C + c];
      sw[c] = t * w_scale;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      float t;
      if (w_nparts == 0) t = w[(size_t)b * C + c];
      else {
        float unused;
        in_partial_sums(w + ((size_t)b * w_nparts * 2) * C + c, w_nparts, C, t, unused);
      }
      sw[c] = t * w_scale;
    }
  }
  __syncthreads();
  const int c4 = C >> 2;                    // threads per pixel
  const int ppb = 256 / c4;                 // pixels per block iteration
  const int sub = threadIdx.x % c4, pl = threadIdx.x / c4;
  for (int p = bx * ppb + pl; p < P; p += gdx * ppb) {
    const float4 v = *reinterpret_cast<const float4*>(A + ((size_t)b * P + p) * C + sub * 4);
    float s = v.x * sw[sub * 4] + v.y * sw[sub * 4 + 1] + v.z * sw[sub * 4 + 2] + v.w * sw[sub * 4 + 3];
    for (int o = c4 >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (sub == 0) cam[(size_t)b * P + p] = fmaxf(s, 0.f);
  }
}

__global__ void __launch_bounds__(256) cam_kernel(const float* __restrict__ A, const float* __restrict__ w,
                                                  int w_nparts, float w_scale, int P, int C,
                                                  float* __restrict__ cam) {
  extern __shared__ float sw[];
  cam_body(A, w, w_nparts, w_scale, P, C, cam, sw, blockIdx.y, blockIdx.x, gridDim.x);
}

// the three Grad-CAM maps of a sweep (grad_cam.layer x3) in one launch: blockIdx.z picks the map
struct CamJob { const float* A; const float* w; float* cam; int nparts, P, C, gx; float scale; };
struct CamJobs { CamJob j[3]; };
__global__ void __launch_bounds__(256) cam3_kernel(const CamJobs js) {
  extern __shared__ float sw[];
  const CamJob& j = js.j[blockIdx.z];
  if ((int)blockIdx.x >= j.gx) return;
  cam_body(j.A, j.w, j.nparts, j.scale, j.P, j.C, j.cam, sw, blockIdx.y, blockIdx.x, j.gx);
}

__device__ __forceinline__ float bilinear_1ch(const float* __restrict__ src, int h, int w, int oy, int ox, int OH,
                                              int OW) {
  const float sy = (oy + 0.5f) * ((float)h / (float)OH) - 0.5f, sx = (ox + 0.5f) * ((float)w / (float)OW) - 0.5f;
  const float fy = floorf(sy), fx = floorf(sx);
  const int ylo = max((int)fy, 0), yhi = min((int)ceilf(sy), h - 1);
  const int xlo = max((int)fx, 0), xhi = min((int)ceilf(sx), w - 1);
  const float ly = sy - fy, lx = sx - fx;
  const float tl = src[ylo * w + xlo], tr = src[ylo * w + xhi], bl = src[yhi * w + xlo], br = src[yhi * w + xhi];
  const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
  return top + (bot - top) * ly;
}

// plz = concat(ldr, cam1, resize(cam2), resize(cam3))  (generator.py:161-164)
__global__ void plz_kernel(const float* __restrict__ ldr, const float* __restrict__ cam1,
                           const float* __restrict__ cam2, const float* __restrict__ cam3, int B, int H, int W,
                           float* __restrict__ plz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H * W) return;
  const int b = i / (H * W), p = i % (H * W), oy = p / W, ox = p % W;
  float* o = plz + (size_t)i * 6;
  o[0] = ldr[(size_t)i * 3]; o[1] = ldr[(size_t)i * 3 + 1]; o[2] = ldr[(size_t)i * 3 + 2];
  o[3] = cam1[i];
  o[4] = bilinear_1ch(cam2 + (size_t)b * (H / 2) * (W / 2), H / 2, W / 2, oy, ox, H, W);
  o[5] = bilinear_1ch(cam3 + (size_t)b * (H / 4) * (W / 4), H / 4, W / 4, oy, ox, H, W);
}

// gamma/beta heads of sunRadNet (sunrad_net.py:52-59), stage 1: flat = leaky(x*scale[c]+shift[c], slope);
// each block reduces one slice of the two Dense(1) dot products -> part[b][slice][2] (fixed order: deterministic).
__global__ void __launch_bounds__(256) dense_heads_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope, int F, int C,
                                                          const float* __restrict__ kg, const float* __restrict__ kb,
                                                          int S, float* __restrict__ part) {
  __shared__ float sred[2][4];
  const int b = blockIdx.x / S, sl = blockIdx.x % S;
  const int per = (F / 4 + S - 1) / S;  // float4 items per slice
  const int beg = sl * per, end = min(F / 4, beg + per);
  float sg = 0.f, sb = 0.f;
  const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)b * F);
  const float4* g4 = reinterpret_cast<const float4*>(kg);
  const float4* b4 = reinterpret_cast<const float4*>(kb);
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    const int c = (i * 4) % C;
    float4 v = x4[i];
    if (scale) {
      v.x = v.x * scale[c] + shift[c]; v.y = v.y * scale[c + 1] + shift[c + 1];
      v.z = v.z * scale[c + 2] + shift[c + 2]; v.w = v.w * scale[c + 3] + shift[c + 3];
    }
    v.x = leaky(v.x, slope); v.y = leaky(v.y, slope); v.z = leaky(v.z, slope); v.w = leaky(v.w, slope);
    const float4 wg = g4[i], wb = b4[i];
    sg += v.x * wg.x + v.y * wg.y + v.z * wg.z + v.w * wg.w;
    sb += v.x * wb.x + v.y * wb.y + v.z * wb.z + v.w * wb.w;
  }
  sg = wave_sum(sg); sb = wave_sum(sb);
  if ((threadIdx.x & 63) == 0) { sred[0][threadIdx.x >> 6] = sg; sred[1][threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)b * S + sl) * 2 + 0] = (sred[0][0] + sred[0][1]) + (sred[0][2] + sred[0][3]);
    part[((size_t)b * S + sl) * 2 + 1] = (sred[1][0] + sred[1][1]) + (sred[1][2] + sred[1][3]);
  }
}

// stage 2 + Dirac-delta sun radiance (sunrad_net.py:54-69 + generator.py:160,167 + tf_utils.py:263-271):
//   gamma/beta = sigmoid(sum_slices part + bias)
//   x = cmf / max(cmf); rad = min(gamma*exp(-(1-x)^2/(beta+1e-5)) / (beta*sqrt(pi)+1e-5), 30000)
// writes rad tiled to 3 channels and its log-compressed (gamma-domain) image.
// *gmax_bits = max(*gmax_bits, bits(max_i x[i])) for NON-NEGATIVE x (the bit pattern of a non-negative float orders like
// the float: an order-independent integer atomicMax) - tf.reduce_max(sunpose_pred) (generator.py:160) when the sun-position
// map is an input of the step instead of the soft-max head's output
__global__ void __launch_bounds__(256) global_max_kernel(const float* __restrict__ x, size_t n, unsigned int* gmax_bits) {
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, x[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(gmax_bits, __float_as_uint(m));
}

__global__ void sun_rad_kernel(const float* __restrict__ cmf, const unsigned int* __restrict__ gmax_bits,
                               const float* __restrict__ part, int S, const float* __restrict__ bg,
                               const float* __restrict__ bb, int B, int P, float* __restrict__ gamma_out,
                               float* __restrict__ beta_out, float* __restrict__ rad_lin3,
                               float* __restrict__ rad_gamma3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * P) return;
  const int b = i / P;
  float ag = bg[0], ab = bb[0];
  for (int s = 0; s < S; ++s) { ag += part[((size_t)b * S + s) * 2]; ab += part[((size_t)b * S + s) * 2 + 1]; }
  const float g = 1.f / (1.f + expf(-ag)), bt = 1.f / (1.f + expf(-ab));
  if (i % P == 0) { gamma_out[b] = g; beta_out[b] = bt; }
  const float gmax = __uint_as_float(*gmax_bits);
  const float x = cmf[i] / gmax;
  const float d = 1.f - x;
  float r = expf(-(d * d) / (bt + 1e-5f)) * g;
  r = r / (bt * 1.7724539f + 1e-5f);  // float32(sqrt(float32(pi)))
  r = r > 30000.f ? 30000.f : r;
  const float rg = logf(1.f + 10.f * r) / 2.3978953f;  // log(11)
  rad_lin3[(size_t)i * 3] = r; rad_lin3[(size_t)i * 3 + 1] = r; rad_lin3[(size_t)i * 3 + 2] = r;
  rad_gamma3[(size_t)i * 3] = rg; rad_gamma3[(size_t)i * 3 + 1] = rg; rad_gamma3[(size_t)i * 3 + 2] = rg;
}

__device__ __forceinline__ float log_decomp(float x) { return (expf(x * 2.3978953f) - 1.f) / 10.f; }

// alpha mask + blending (inference.py:91-94,109-113 ; train.py:258-261,293-299 ; generator.py:171-175)
__global__ void blend_kernel(const float* __restrict__ sky_gamma, const float* __restrict__ sun_gamma, int npix,
                             float thr, float* __restrict__ y_gamma, float* __restrict__ y_lin,
                             float* __restrict__ alpha, float* __restrict__ sky_lin, float* __restrict__ sun_lin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  float sk[3], su[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { sk[c] = sky_gamma[(size_t)i * 3 + c]; su[c] = sun_gamma[(size_t)i * 3 + c]; }
  const float m = fmaxf(fmaxf(log_decomp(sk[0]), log_decomp(sk[1])), log_decomp(sk[2]));
  const float a = fminf(1.f, fmaxf(0.f, m - 1.f + thr) / thr);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float s = (1.f - a) * sk[c], u = a * su[c];
    const float yg = s + u;
    y_gamma[(size_t)i * 3 + c] = yg;
    y_lin[(size_t)i * 3 + c] = log_decomp(yg);
    if (alpha) alpha[(size_t)i * 3 + c] = a;
    if (sky_lin) sky_lin[(size_t)i * 3 + c] = log_decomp(s);
    if (sun_lin) sun_lin[(size_t)i * 3 + c] = log_decomp(u);
  }
}

// tf_utils.hdr_logCompression / hdr_logDecompression (tf_utils.py:263-280), validDR = 10
__global__ void tonemap_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int decompress) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = decompress ? log_decomp(v) : logf(1.f + 10.f * v) / 2.3978953f;
  }
}

// ops.relu (ops.py:324-329) / LeakyReLU as a standalone layer: y = x > 0 ? x : slope * x
__global__ void leaky_relu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float slope) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = v > 0.f ? v : slope * v;
  }
}

// train.py:54-94 `_preprocessing` without the JPEG round trip: exposure, signal-dependent + constant Gaussian noise,
// relu, clip to [0,1], camera response function by linear interpolation of a K-sample LUT (tf_utils.apply_rf /
// interp_1d / sample_1d, tf_utils.py:191-255: index clipped to [0, K-1]), 8-bit quantisation with round-half-to-even.
__global__ void ldr_synth_kernel(const float* __restrict__ hdr, const float* __restrict__ t, const float* __restrict__ sigma_s,
                                 const float* __restrict__ sigma_c, const float* __restrict__ noise_s,
                                 const float* __restrict__ noise_c, const float* __restrict__ crf, int K, size_t per_sample,
                                 size_t total, float* __restrict__ hdr_t, float* __restrict__ ldr) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_sample), c = (int)(i % 3);
    const float x0 = hdr[i] * t[b];
    const float ns = noise_s[i] * (sigma_s[b * 3 + c] * x0);
    float x = x0 + ns;
    x = x + sigma_c[b * 3 + c] * noise_c[i];
    x = fmaxf(x, 0.f);
    hdr_t[i] = x;
    const float cl = fminf(fmaxf(x, 0.f), 1.f);
    const float pos = (float)(K - 1) * cl;
    const float y0 = floorf(pos), y1 = y0 + 1.f;
    const int i0 = min(max((int)y0, 0), K - 1), i1 = min(max((int)y1, 0), K - 1);
    const float* lut = crf + (size_t)b * K;
    const float v = (y1 - pos) * lut[i0] + (pos - y0) * lut[i1];
    ldr[i] = rintf(v * 255.f) / 255.f;
  }
}

// train.py:42-52 `vMF`: pmf[b][j] = exp(kappa * <bin_j, sun_b>) / sum_j, bins = tf_utils.sunpose_init (tf_utils.py:112-129),
// sun_b = tf_utils.sphere2world((azimuth, elevation_b), h, w, skydome=True) (tf_utils.py:95-110).  One block per sample.
__global__ void __launch_bounds__(256) vmf_target_kernel(const float* __restrict__ elevation, float azimuth, int h, int w,
                                                         float kappa, float* __restrict__ out) {
  __shared__ float sred[4];
  const int b = blockIdx.x, n = h * w;
  const float pi = 3.14159265358979323846f;
  const float th_s = (azimuth - 0.5f * (float)w) * (2.f * pi / (float)w);
  const float ph_s = ((float)h - elevation[b]) * (pi / (float)(h * 2));
  const float sx = cosf(ph_s) * cosf(th_s), sy = sinf(ph_s), sz = cosf(ph_s) * sinf(th_s);
  float part = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float row = floorf((float)j / (float)w);
    const float xd = (((float)j + 1.f) - row * (float)w - 1.f) * (360.f / (float)w) + (360.f / ((float)w * 2.f));
    const float yd = row * (90.f / (float)h) + (90.f / (2.f * (float)h));
    const float phi = yd * (pi / 180.f), theta = (xd - 180.f) * (pi / 180.f);
    const float d = cosf(phi) * cosf(theta) * sx + sinf(phi) * sy + cosf(phi) * sinf(theta) * sz;
    const float e = expf(kappa * d);
    out[(size_t)b * n + j] = e;
    part += e;
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = part;
  __syncthreads();
  const float tot = sred[0] + sred[1] + sred[2] + sred[3];
  for (int j = threadIdx.x; j < n; j += 256) out[(size_t)b * n + j] /= tot;
}

// Debug aid: workgroups that sit on the CUs with a known LDS pattern and report words that change under them
// (= some co-resident workgroup of another kernel wrote outside its own LDS allocation).  Bounded loop.
__global__ void __launch_bounds__(64) lds_canary_kernel(int iters, unsigned int* report) {
  __shared__ unsigned int pat[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) pat[i] = 0xC0FFEE00u ^ (unsigned)i;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_sleep(64);
    for (int i = threadIdx.x; i < 1024; i += 64) {
      const unsigned int v = pat[i];
      if (v != (0xC0FFEE00u ^ (unsigned)i)) {
        const unsigned int n = atomicAdd(report, 1u);
        if (n < 15) { report[1 + 2 * n] = (unsigned)i; report[2 + 2 * n] = v; }
        pat[i] = 0xC0FFEE00u ^ (unsigned)i;
      }
    }
  }
}

}  // namespace

extern "C" {

int hdrsky_ldr_synth(const float* hdr, const float* t, const float* sigma_s, const float* sigma_c, const float* noise_s,
                     const float* noise_c, const float* crf, int crf_len, int B, int H, int W, float* hdr_t, float* ldr,
                     void* stream) {
  if (!hdr || !t || !sigma_s || !sigma_c || !noise_s || !noise_c || !crf || !hdr_t || !ldr || crf_len < 2) return HDRSKY_EINVAL;
  const size_t per = (size_t)H * W * 3, total = per * B;
  size_t g = (total + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
  hipLaunchKernelGGL(ldr_synth_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, hdr, t, sigma_s, sigma_c, noise_s,
                     noise_c, crf, crf_len, per, total, hdr_t, ldr);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_vmf_target(const float* elevation, float azimuth, int B, int H, int W, float kappa, float* out, void* stream) {
  if (!elevation || !out || B <= 0) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(vmf_target_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, elevation, azimuth, H, W, kappa, out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_debug_lds_canary(int nblocks, int iters, void* report, void* stream) {
  hipLaunchKernelGGL(lds_canary_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, iters, (unsigned int*)report);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_norm_apply(const float* x, int x_bf16, const float* part, int nparts, const float* gamma, const float* beta, float eps,
                      float slope, const float* residual, float* y, float* ypool, int B, int H, int W, int C,
                      void* stream) {
  if (!x || !part || !gamma || !beta || !y || (C & 3) || C > 1024) return HDRSKY_EINVAL;
  if (ypool && ((H | W) & 1)) return HDRSKY_EINVAL;
  int S = 256 / B; if (S < 1) S = 1; if (S > 64) S = 64;
  hipLaunchKernelGGL(norm_apply_kernel, dim3(B * S), dim3(256), 2 * C * sizeof(float), (hipStream_t)stream, x, part,
                     nparts, gamma, beta, eps, slope, residual, y, ypool, B, H, W, C, S, x_bf16 ? 1 : 0);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_in_finalize(const float* part, int nparts, int B, int C, int count, const float* gamma, const float* beta,
                       float eps, float* mean, float* rstd, float* scale, float* shift, void* stream) {
  if (!part || !gamma || !beta) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(in_finalize_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, part, nparts, B, C,
                     1.f / (float)count, gamma, beta, eps, mean, rstd, scale, shift);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_in_affine(const float* part, int nparts, int B, int C, int count, const float* gamma, const float* beta, float eps,
                     float* scale, float* shift, void* stream) {
  if (!part || !gamma || !beta || !scale || !shift || nparts <= 0 || count <= 0) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(in_affine_kernel, dim3(B * cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, part, nparts, B, C,
                     1.f / (float)count, gamma, beta, eps, scale, shift);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

// hdrsky_in_affine for a PAIRED tensor: samples [0, B/2) are one layer's output (gamma, beta), samples [B/2, B) another's
// (gamma2, beta2) - the same values as two calls on the halves
int hdrsky_in_affine_pair(const float* part, int nparts, int B, int C, int count, const float* gamma, const float* beta,
                          const float* gamma2, const float* beta2, float eps, float* scale, float* shift, void* stream) {
  if (!part || !gamma || !beta || !gamma2 || !beta2 || !scale || !shift || nparts <= 0 || count <= 0 || (B & 1)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(in_affine_kernel, dim3(B * cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, part, nparts, B, C,
                     1.f / (float)count, gamma, beta, eps, scale, shift, gamma2, beta2, B / 2);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_bn_eval_affine(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                          float eps, int C, float* scale, float* shift, void* stream) {
  if (!gamma || !beta || !moving_mean || !moving_var || !scale || !shift) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     moving_mean, moving_var, eps, C, scale, shift);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

// spatial slices per (sample, 16-channel group): aim for >= 512 workgroups, at least 256 pixels (windows) each
int hdrsky_norm_act_bwd_nslices(int B, int H, int W, int C, int pooled) {
  const int units = pooled ? (H / 2) * (W / 2) : H * W;
  const int groups = B * (C / 16);
  const int target = hdrsky_hooks().nab_target;   // (tuning hook; 512)
  int S = groups > 0 ? (target + groups - 1) / groups : 1;
  const int smax = units / 256 > 0 ? units / 256 : 1;   // below ~256 pixels per slice the second launch costs more
  if (S > smax) S = smax;
  if (S > 64) S = 64;
  return S < 1 ? 1 : S;
}

// Shapes the one-launch form takes (pixels - or 2x2 windows of the pooled form - per sample: the maps of the 32x128 network
// and the 32x128 map of the 128x512 one).  HDRSKY_NAB_ONE=0: the sliced form (switch).
int hdrsky_norm_act_bwd_one_launch(int H, int W, int pooled, int dy_bf16) {
  (void)dy_bf16;
  const int mode = hdrsky_hooks().nab_one;       // 1: every supported shape; 2 / 3 (A/B): the non-pooled / the pooled forms only
  if (!mode || (mode == 2 && pooled) || (mode == 3 && !pooled)) return 0;
  const int units = pooled ? (H / 2) * (W / 2) : H * W;
  if (pooled) return units == 1024 || units == 256;
  // (the 4096-pixel maps keep the sliced pair: their slab needs 8-channel groups to fit the registers, i.e. 32 B of every 128-B line
  // per workgroup - 22.4 us alone against 17.9 us for the two sliced launches, profiles/r05_microbench_nab.txt; a 16-channel slab
  // spills: 29.6 us)
  return units == 1024 || units == 256;
}

static int norm_act_bwd_impl(const float* x, const float* part, int nparts, const float* gamma, const float* beta,
                             float eps, float slope, const float* dy, int pooled, void* dx, int dx_bf16, float* sums,
                             float* dgamma, float* dbeta, float* ws, int B, int H, int W, int C, void* stream,
                             const float* gamma2, const float* beta2, int gsplit) {
  if (!x || !part || !gamma || !beta || !dy || !dx || (C & 15)) return HDRSKY_EINVAL;
  if (pooled && ((H | W) & 1)) return HDRSKY_EINVAL;
  // (a paired tensor is sliced like ONE of its layers' launches: the same partial sums in the same order, bit for bit)
  const int S = hdrsky_norm_act_bwd_nslices(gsplit > 0 ? gsplit : B, H, W, C, pooled);
  const int groups = B * (C / 16);
  // the register-resident single launch (norm_act_bwd1_kernel) where a (sample, channel group) slab is NT x NV units
  // (calls that store dx as bf16 - the single-product mode's: the fp32-class mode keeps the sliced form, so that its training
  // trajectories - bench.py's parity fit - stay those of round 4 bit for bit)
  if ((dx_bf16 & 1) && hdrsky_norm_act_bwd_one_launch(H, W, pooled, (dx_bf16 & 2) != 0)) {
    const int units = pooled ? (H / 2) * (W / 2) : H * W;
    const bool dy16 = (dx_bf16 & 2) != 0;
    if (hdrsky_hooks().nab_nt) dx_bf16 |= 8;
#define HDRSKY_NAB1(NT_, NV_, CG_, P_)                                                                                              \
    do {                                                                                                                             \
      if ((C % CG_) != 0) return HDRSKY_EINVAL;                                                                                      \
      if (dy16)                                                                                                                      \
        hipLaunchKernelGGL((norm_act_bwd1_kernel<NT_, NV_, CG_, P_, true>), dim3(B * (C / CG_)), dim3(NT_), 0, (hipStream_t)stream, x, \
                           part, nparts, gamma, beta, eps, slope, dy, dx, dx_bf16, sums, dgamma, dbeta, H, W, C, gamma2, beta2, gsplit); \
      else                                                                                                                           \
        hipLaunchKernelGGL((norm_act_bwd1_kernel<NT_, NV_, CG_, P_, false>), dim3(B * (C / CG_)), dim3(NT_), 0, (hipStream_t)stream, x, \
                           part, nparts, gamma, beta, eps, slope, dy, dx, dx_bf16, sums, dgamma, dbeta, H, W, C, gamma2, beta2, gsplit); \
    } while (0)
    // units per sample -> (threads, units per thread, channels per workgroup): 32 values of xhat per thread at most - the pooled
    // form of the 4096-pixel maps on 8-channel groups
    if (!pooled && units == 1024) HDRSKY_NAB1(1024, 4, 16, false);
    else if (!pooled && units == 256) HDRSKY_NAB1(256, 4, 16, false);
    else if (pooled && units == 1024) HDRSKY_NAB1(1024, 2, 8, true);
    else if (pooled && units == 256) HDRSKY_NAB1(512, 2, 16, true);
    else return HDRSKY_EINVAL;
#undef HDRSKY_NAB1
    HDRSKY_CHECK_LAUNCH();
    return HDRSKY_OK;
  }
  if (S == 1) {
    hipLaunchKernelGGL(norm_act_bwd_kernel<2>, dim3(groups), dim3(256), 0, (hipStream_t)stream, x, part, nparts, gamma,
                       beta, eps, slope, dy, pooled, dx, dx_bf16, sums, dgamma, dbeta, ws, 1, B, H, W, C, gamma2, beta2, gsplit);
    HDRSKY_CHECK_LAUNCH();
    return HDRSKY_OK;
  }
  if (!ws) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(norm_act_bwd_kernel<0>, dim3(groups * S), dim3(256), 0, (hipStream_t)stream, x, part, nparts, gamma,
                     beta, eps, slope, dy, pooled, dx, dx_bf16, sums, dgamma, dbeta, ws, S, B, H, W, C, gamma2, beta2, gsplit);
  HDRSKY_CHECK_LAUNCH();
  hipLaunchKernelGGL(norm_act_bwd_kernel<1>, dim3(groups * S), dim3(256), 0, (hipStream_t)stream, x, part, nparts, gamma,
                     beta, eps, slope, dy, pooled, dx, dx_bf16, sums, dgamma, dbeta, ws, S, B, H, W, C, gamma2, beta2, gsplit);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_norm_act_bwd(const float* x, const float* part, int nparts, const float* gamma, const float* beta,
                        float eps, float slope, const float* dy, int pooled, void* dx, int dx_bf16, float* sums,
                        float* dgamma, float* dbeta, float* ws, int B, int H, int W, int C, void* stream) {
  return norm_act_bwd_impl(x, part, nparts, gamma, beta, eps, slope, dy, pooled, dx, dx_bf16, sums, dgamma, dbeta, ws, B, H, W, C, stream,
                           nullptr, nullptr, 0);
}

// hdrsky_norm_act_bwd on a PAIRED tensor: samples [0, B/2) belong to the layer with (gamma, beta), samples [B/2, B) to the one with
// (gamma2, beta2) - one launch, the values of two calls on the halves (the statistics, sums and the formula are per sample; the
// workspace ws is that of a B-sample call).  No atomics form: the per-sample terms go to `sums`.
int hdrsky_norm_act_bwd_pair(const float* x, const float* part, int nparts, const float* gamma, const float* beta, const float* gamma2,
                             const float* beta2, float eps, float slope, const float* dy, int pooled, void* dx, int dx_bf16,
                             float* sums, float* ws, int B, int H, int W, int C, void* stream) {
  if (!gamma2 || !beta2 || (B & 1)) return HDRSKY_EINVAL;
  return norm_act_bwd_impl(x, part, nparts, gamma, beta, eps, slope, dy, pooled, dx, dx_bf16, sums, nullptr, nullptr, ws, B, H, W, C, stream,
                           gamma2, beta2, B / 2);
}

int hdrsky_global_max(const float* x, size_t n, void* gmax_bits, void* stream) {
  if (!x || !gmax_bits || n == 0) return HDRSKY_EINVAL;
  const size_t blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(global_max_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, (hipStream_t)stream, x, n,
                     (unsigned int*)gmax_bits);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_softmax_head(const float* part, int nsplit, int M, int N, const float* bias, float* z, float* cmf,
                        void* gmax_bits, void* stream) {
  if (!part || !cmf) return HDRSKY_EINVAL;
  if ((N & 3) || N > 4 * 4 * 1024) return HDRSKY_EUNSUPPORTED;   // a row lives in the registers of one 1024-thread block
  if (N <= 4096)
    hipLaunchKernelGGL((softmax_head_kernel<1, false>), dim3(M), dim3(1024), 0, (hipStream_t)stream, part, nsplit, M, N, bias, z,
                       cmf, (unsigned int*)gmax_bits, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((softmax_head_kernel<4, false>), dim3(M), dim3(1024), 0, (hipStream_t)stream, part, nsplit, M, N, bias, z,
                       cmf, (unsigned int*)gmax_bits, nullptr, nullptr, nullptr);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_softmax_head_pick(const float* part, int nsplit, int M, int N, const float* bias, float* z, float* cmf,
                             void* gmax_bits, const float* pick_src, float* dz, int* idx_out, void* stream) {
  if (!part || !cmf || !dz) return HDRSKY_EINVAL;
  if ((N & 3) || N > 4 * 4 * 1024) return HDRSKY_EUNSUPPORTED;
  if (N <= 4096)
    hipLaunchKernelGGL((softmax_head_kernel<1, true>), dim3(M), dim3(1024), 0, (hipStream_t)stream, part, nsplit, M, N, bias, z,
                       cmf, (unsigned int*)gmax_bits, pick_src, dz, idx_out);
  else
    hipLaunchKernelGGL((softmax_head_kernel<4, true>), dim3(M), dim3(1024), 0, (hipStream_t)stream, part, nsplit, M, N, bias, z,
                       cmf, (unsigned int*)gmax_bits, pick_src, dz, idx_out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_softmax_pick_bwd(const float* cmf, const float* z, const float* pick_src, int M, int N, float* dz,
                            int* idx_out, void* stream) {
  if (!cmf || !z || !pick_src || !dz) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(softmax_pick_bwd_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, cmf, z, pick_src, N, dz,
                     idx_out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_fc_finalize(const float* part, int nsplit, int M, int N, const float* bias, int relu, const float* mask_src,
                       float* y, void* zero_word, void* stream) {
  if (!part || !y) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(fc_finalize_kernel, dim3(cdiv(M * N, 256)), dim3(256), 0, (hipStream_t)stream, part, nsplit, M, N,
                     bias, relu, mask_src, y, (unsigned int*)zero_word);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_spatial_sum(const float* x, int B, int P, int C, float scale, float* out, void* stream) {
  if (!x || !out || C > 256 || (256 % C) != 0) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(spatial_sum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, P, C, scale, out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_grad_cam(const float* A, const float* w, int w_nparts, float w_scale, int B, int P, int C, float* cam,
                    void* stream) {
  if (!A || !w || !cam || (C & 3) || C > 256 || (256 % (C / 4)) != 0 || w_nparts < -256) return HDRSKY_EINVAL;
  const int ppb = 256 / (C / 4);
  int gx = cdiv(P, ppb); if (gx > 64) gx = 64;
  const size_t lds = (C + (w_nparts < 0 ? 1024 : 0)) * sizeof(float);     // + [256 / (C/4) slices][C] partial sums
  hipLaunchKernelGGL(cam_kernel, dim3(gx, B), dim3(256), lds, (hipStream_t)stream, A, w, w_nparts, w_scale, P, C, cam);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_grad_cam3(const float* const* A, const float* const* w, const int* w_nparts, const float* w_scale, const int* P,
                     const int* C, float* const* cam, int B, void* stream) {
  if (!A || !w || !w_nparts || !w_scale || !P || !C || !cam || B <= 0) return HDRSKY_EINVAL;
  CamJobs js;
  int gmax = 1;
  size_t lds = 0;
  for (int k = 0; k < 3; ++k) {
    if (!A[k] || !w[k] || !cam[k] || (C[k] & 3) || C[k] > 256 || (256 % (C[k] / 4)) != 0 || w_nparts[k] < -256) return HDRSKY_EINVAL;
    const int ppb = 256 / (C[k] / 4);
    int gx = cdiv(P[k], ppb); if (gx > 64) gx = 64;
    js.j[k] = CamJob{A[k], w[k], cam[k], w_nparts[k], P[k], C[k], gx, w_scale[k]};
    if (gx > gmax) gmax = gx;
    const size_t need = (C[k] + (w_nparts[k] < 0 ? 1024 : 0)) * sizeof(float);
    if (need > lds) lds = need;
  }
  hipLaunchKernelGGL(cam3_kernel, dim3(gmax, B, 3), dim3(256), lds, (hipStream_t)stream, js);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_plz_build(const float* ldr, const float* cam1, const float* cam2, const float* cam3, int B, int H, int W,
                     float* plz, void* stream) {
  if (!ldr || !cam1 || !cam2 || !cam3 || !plz || (H & 3) || (W & 3)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(plz_kernel, dim3(cdiv(B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, ldr, cam1, cam2, cam3,
                     B, H, W, plz);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_dense_heads(const float* x, const float* scale, const float* shift, float slope, int B, int F, int C,
                       const float* kg, const float* kb, int S, float* part, void* stream) {
  if (!x || !kg || !kb || !part || S <= 0 || (F & 3) || (C & 3)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(dense_heads_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, x, scale, shift, slope, F, C, kg,
                     kb, S, part);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_sun_rad(const float* cmf, const void* gmax_bits, const float* part, int S, const float* bg, const float* bb,
                   int B, int P, float* gamma_out, float* beta_out, float* rad_lin3, float* rad_gamma3, void* stream) {
  if (!cmf || !gmax_bits || !part || !bg || !bb || !gamma_out || !beta_out || !rad_lin3 || !rad_gamma3) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(sun_rad_kernel, dim3(cdiv(B * P, 256)), dim3(256), 0, (hipStream_t)stream, cmf,
                     (const unsigned int*)gmax_bits, part, S, bg, bb, B, P, gamma_out, beta_out, rad_lin3, rad_gamma3);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_blend(const float* sky_gamma, const float* sun_gamma, int npix, float thr, float* y_gamma, float* y_lin,
                 float* alpha, float* sky_lin, float* sun_lin, void* stream) {
  if (!sky_gamma || !sun_gamma || !y_gamma || !y_lin) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(blend_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, sky_gamma, sun_gamma, npix,
                     thr, y_gamma, y_lin, alpha, sky_lin, sun_lin);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_tonemap(const float* x, float* y, size_t n, int decompress, void* stream) {
  if (!x || !y) return HDRSKY_EINVAL;
  size_t g = (n + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
  hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, n, decompress);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_leaky_relu(const float* x, float* y, size_t n, float slope, void* stream) {
  if (!x || !y) return HDRSKY_EINVAL;
  size_t g = (n + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
  hipLaunchKernelGGL(leaky_relu_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, n, slope);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

}  // extern "C"
