// Shared device/host helpers for libhdrsky (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hdrsky.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define HDRSKY_CHECK_LAUNCH()                                  \
  do {                                                         \
    if (hipGetLastError() != hipSuccess) return HDRSKY_ELAUNCH; \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int roundup(int a, int b) { return cdiv(a, b) * b; }

// fp32 -> bf16 round-to-nearest-even (v_cvt_pk_bf16_f32) and its fp32 residual.
__device__ __forceinline__ unsigned short f2bf(float x) {
  __bf16 h = (__bf16)x;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned short u) {
  unsigned int v = ((unsigned int)u) << 16;
  return __builtin_bit_cast(float, v);
}

// Packs 8 floats into 8 bf16 (hi) and optionally the bf16 of the residuals (lo).
template <bool PRECISE>
__device__ __forceinline__ void pack8(const float (&v)[8], uint4& hi, uint4& lo) {
  unsigned short h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h[j] = f2bf(v[j]);
    if (PRECISE) l[j] = f2bf(v[j] - bf2f(h[j]));
  }
  hi.x = h[0] | ((unsigned)h[1] << 16);
  hi.y = h[2] | ((unsigned)h[3] << 16);
  hi.z = h[4] | ((unsigned)h[5] << 16);
  hi.w = h[6] | ((unsigned)h[7] << 16);
  if (PRECISE) {
    lo.x = l[0] | ((unsigned)l[1] << 16);
    lo.y = l[2] | ((unsigned)l[3] << 16);
    lo.z = l[4] | ((unsigned)l[5] << 16);
    lo.w = l[6] | ((unsigned)l[7] << 16);
  }
}

__device__ __forceinline__ float leaky(float v, float slope) { return v >= 0.f ? v : v * slope; }

// Four / eight consecutive elements (element offset e, a multiple of 4 / 8) of a tensor stored as fp32 or as bf16: gradients
// with one reader (hdrsky_conv_desc.y_bf16 of a data-gradient conv) and - round 4 - raw conv outputs in front of a norm layer
// travel as bf16 in the single-product mode; the readers widen them on the way in.
__device__ __forceinline__ float4 ld4any(const float* p, int is_bf16, size_t e) {
  if (is_bf16) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + e);
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(p + e);
}
__device__ __forceinline__ void ld8any(const float* p, int is_bf16, size_t e, float4& lo, float4& hi) {
  if (is_bf16) {
    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + e);
    lo = make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
    hi = make_float4(__builtin_bit_cast(float, u.z << 16), __builtin_bit_cast(float, u.z & 0xffff0000u),
                     __builtin_bit_cast(float, u.w << 16), __builtin_bit_cast(float, u.w & 0xffff0000u));
  } else {
    lo = *reinterpret_cast<const float4*>(p + e); hi = *reinterpret_cast<const float4*>(p + e + 4);
  }
}

// (sum, sum of squares) of one channel over the nparts tile partials of a sample: pp -> the channel's sum in tile 0, tiles
// 2*C floats apart, the squares C floats behind the sums.  Loads go out eight at a time (independent, all in flight), the
// additions stay in tile order: bit-identical to the plain loop, without its chain of nparts dependent loads - which sat
// in the prologue of every kernel that normalises on the fly (64 tiles at 32x128: ~10 us).  Adding in interleaved chains
// instead would move the statistics by an ulp - enough to flip max-pool arg-max ties in the Grad-CAM sweep.
__device__ __forceinline__ void in_partial_sums(const float* __restrict__ pp, int nparts, int C, float& s, float& ss) {
  s = 0.f; ss = 0.f;
  int p = 0;
  for (; p + 7 < nparts; p += 8) {
    float a8[8], q8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a8[k] = pp[(size_t)(2 * (p + k)) * C]; q8[k] = pp[(size_t)(2 * (p + k) + 1) * C]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += a8[k]; ss += q8[k]; }
  }
  for (; p < nparts; ++p) { s += pp[(size_t)(2 * p) * C]; ss += pp[(size_t)(2 * p + 1) * C]; }
}

__device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                 c, 0, 0, 0);
}

// Wave-level sum over the 64 lanes (result in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// one bilinear sample position of the reference (distortion_aware_ops.py:62-106), all in float32
struct Tap4 {
  int y0, y1, x0, x1;      // corner indices in PADDED coordinates (x wrapped, y clamped)
  float w0, w1, w2, w3;
};

// The sample position is separable: the row part depends on (image row, tap) only, the column part on (column, tap).
// da_tap composes the two halves, so a kernel that computes them separately (the row half once per row and tap) gets
// bit-identical corners and weights.
struct TapAxis {
  int i0, i1;        // corner indices in PADDED coordinates
  float wa, wb;      // (f(i1) - pos) and (pos - f(i0)): the weights of i0 and i1 along this axis
};

__host__ __device__ __forceinline__ TapAxis da_tap_y(float base_y, float off_y, int in_h) {
  float y = base_y + off_y;
  y = y < 0.f ? 0.f : y; y = y > (float)(in_h - 1) ? (float)(in_h - 1) : y;
  int y0 = (int)floorf(y);
  int y1 = y0 + 1;
  y0 = y0 < 0 ? 0 : (y0 > in_h - 1 ? in_h - 1 : y0);
  y1 = y1 < 0 ? 0 : (y1 > in_h - 1 ? in_h - 1 : y1);
  TapAxis t;
  t.i0 = y0; t.i1 = y1;
  t.wa = (float)y1 - y; t.wb = y - (float)y0;
  return t;
}

__host__ __device__ __forceinline__ TapAxis da_tap_x(float base_x, float off_x, int in_w) {
  float x = base_x + off_x;
  x = x < 0.f ? x + (float)in_w : x;
  x = x > (float)(in_w - 1) ? x - (float)in_w : x;
  int x0 = (int)floorf(x);
  int x1 = x0 + 1;
  const int x0w = x0, x1w = x1;  // unwrapped: used for the weights (:89, :100-106)
  x0 = x0 < 0 ? x0 + in_w : x0; x1 = x1 < 0 ? x1 + in_w : x1;
  x0 = x0 > in_w - 1 ? x0 - in_w : x0; x1 = x1 > in_w - 1 ? x1 - in_w : x1;
  TapAxis t;
  t.i0 = x0; t.i1 = x1;
  t.wa = (float)x1w - x; t.wb = x - (float)x0w;
  return t;
}

__host__ __device__ __forceinline__ Tap4 da_tap(float base_y, float base_x, float off_y, float off_x, int in_h, int in_w) {
  const TapAxis ty = da_tap_y(base_y, off_y, in_h), tx = da_tap_x(base_x, off_x, in_w);
  Tap4 t;
  t.y0 = ty.i0; t.y1 = ty.i1; t.x0 = tx.i0; t.x1 = tx.i1;
  t.w0 = ty.wa * tx.wa;
  t.w1 = ty.wa * tx.wb;
  t.w2 = ty.wb * tx.wa;
  t.w3 = ty.wb * tx.wb;
  return t;
}

// LDS-DMA of 16 bytes per lane: LDS destination = lds_dst (wave-uniform byte address) + lane * 16.  Inline asm so that
// the compiler's vmcnt bookkeeping does not see it (it would drain every DMA before the first ds_read): completion is
// waited for by the explicit counted s_waitcnt below.  M0 is saved / restored inside the statement.
// ... with the non-temporal policy (bytes one workgroup reads once and nobody re-reads soon)
__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
