// Dense weight gradient on the matrix cores, and the RMSprop update that never materialises it.
//
//   g[k][n] = sum_m x[m][k] * dy[m][n]          (tf.gradients through Keras Dense, sunpose_net.py:48-51,65-68;
//                                                optimizer: train.py:403 RMSprop(lr), rho 0.9, eps 1e-7)
//
// The two sun-pose Dense kernels hold 50.3 M of the 58.3 M trainables.  Written out, their gradient costs 168 MB of
// HBM writes and 168 MB of reads per step beside the 840 MB the update itself moves; the contraction behind it is
// 2*M flops per weight (M = batch rows, 32..256) - free next to that traffic.  So the update kernel recomputes the
// gradient tile it is about to apply:  workgroup = 128 k x 128 n of the kernel, x[:, k-tile] and dy[:, n-tile] are
// staged once per 32 rows as bf16 (transposed, so that a lane's 8 consecutive m are one ds_read_b128) and contracted
// with v_mfma_f32_32x32x16_bf16; every lane then owns 64 weights whose w / ms it reads, updates and writes, together
// with the two bf16 MFMA images of the new weights (packed [K/8][N][8] and natural [K][N], see fc.hip).
// HBM-bound: 20 B per weight (w, ms read + written, 2 x 2 B images); algorithmic minimum of an RMSprop step.
//
// The same contraction with a plain store epilogue (hdrsky_fc_wgrad_bf16) is the materialised gradient for callers that
// need it (a data-parallel all-reduce, gradient inspection).  Operands are rounded to bf16 (2^-9 relative per factor,
// fp32 accumulation) - the BF16 compute mode's contract, like every convolution weight gradient; the bias gradient is
// summed from the fp32 rows.  BF16X3 keeps hdrsky_fc_wgrad (fp32 FMA).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int TK = 128, TN = 128;   // kernel tile of one workgroup (4 waves, 2 x 2, 64 x 64 each)
constexpr int MC = 32;              // batch rows per staging round
constexpr int ROWB = 80;            // LDS row: 32 bf16 + 16 B pad (ds_read_b128 of 16 rows then covers all 64 banks)

__device__ __forceinline__ f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                 0, 0);
}

template <bool FUSED>
__global__ void __launch_bounds__(256, FUSED ? 2 : 4) fc_xtdy_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                      int ldy, int M, int K, int N, float* __restrict__ w,
                                                      float* __restrict__ ms, float lr, float rho, float eps,
                                                      float gscale, uint2* __restrict__ pk_hi,
                                                      unsigned short* __restrict__ nat_hi, float* __restrict__ db,
                                                      int accumulate) {
  __shared__ __attribute__((aligned(16))) unsigned char sx[TK * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char sd[TN * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int k0 = blockIdx.y * TK, n0 = blockIdx.x * TN;
  const int wk = (wave >> 1) * 64, wn = (wave & 1) * 64;

  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  for (int m0 = 0; m0 < M; m0 += MC) {
    if (m0) __syncthreads();
    // 16 row pairs x 32 column quads per matrix; a pair of rows becomes one 32-bit LDS word per column
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int item = it * 256 + tid;
      const bool isd = item >= 512;
      const int mp = (item & 511) >> 5, cq = item & 31;
      const int m = m0 + 2 * mp;
      const int ld = isd ? ldy : ldx;
      const float* src = isd ? dy + (size_t)m * ldy + n0 + cq * 4 : x + (size_t)m * ldx + k0 + cq * 4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (m < M) a = *reinterpret_cast<const float4*>(src);
      if (m + 1 < M) b = *reinterpret_cast<const float4*>(src + ld);
      unsigned char* dst = (isd ? sd : sx) + (cq * 4) * ROWB + mp * 4;
      *reinterpret_cast<unsigned*>(dst) = f2bf(a.x) | ((unsigned)f2bf(b.x) << 16);
      *reinterpret_cast<unsigned*>(dst + ROWB) = f2bf(a.y) | ((unsigned)f2bf(b.y) << 16);
      *reinterpret_cast<unsigned*>(dst + 2 * ROWB) = f2bf(a.z) | ((unsigned)f2bf(b.z) << 16);
      *reinterpret_cast<unsigned*>(dst + 3 * ROWB) = f2bf(a.w) | ((unsigned)f2bf(b.w) << 16);
    }
    __syncthreads();
#pragma unroll
    for (int step = 0; step < 2; ++step) {
      uint4 af[2], bf[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        af[q] = *reinterpret_cast<const uint4*>(sx + (wk + q * 32 + r) * ROWB + (step * 2 + h) * 16);
        bf[q] = *reinterpret_cast<const uint4*>(sd + (wn + q * 32 + r) * ROWB + (step * 2 + h) * 16);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(af[a], bf[b], acc[a][b]);
    }
  }

  // bias gradient: fp32 column sums of dy, by the workgroups of the first k tile
  if (db && blockIdx.y == 0 && tid < TN) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[(size_t)m * ldy + n0 + tid];
    db[n0 + tid] = accumulate ? db[n0 + tid] + s : s;
  }

  // D layout of 32x32: register i of lane (r, h) is row (i & 3) + 8 (i >> 2) + 4 h, column r
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kb = k0 + wk + a * 32, n = n0 + wn + b * 32 + r;
      if (FUSED) {
        float wv[16], mv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const size_t idx = (size_t)(kb + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n;
          wv[i] = w[idx]; mv[i] = ms[idx];
        }
        unsigned short hb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const size_t idx = (size_t)(kb + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n;
          const float gg = acc[a][b][i] * gscale;
          const float m_ = rho * mv[i] + (1.f - rho) * gg * gg;
          const float nw = wv[i] - lr * gg / (sqrtf(m_) + eps);
          ms[idx] = m_; w[idx] = nw;
          hb[i] = f2bf(nw);
          if (nat_hi) nat_hi[idx] = hb[i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // rows 8q + 4h .. +3: half of one packed 16-byte group
          const size_t oct = (size_t)((kb >> 3) + q) * N + n;
          pk_hi[oct * 2 + h] = uint2{hb[4 * q] | ((unsigned)hb[4 * q + 1] << 16), hb[4 * q + 2] | ((unsigned)hb[4 * q + 3] << 16)};
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const size_t idx = (size_t)(kb + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n;
          w[idx] = accumulate ? w[idx] + acc[a][b][i] : acc[a][b][i];
        }
      }
    }
  }
}

inline bool shapes_ok(int ldx, int ldy, int M, int K, int N) {
  return M > 0 && K > 0 && N > 0 && (K % TK) == 0 && (N % TN) == 0 && ldx >= K && ldy >= N && (ldx & 3) == 0 && (ldy & 3) == 0;
}

}  // namespace

extern "C" {

int hdrsky_fc_wgrad_bf16(const float* x, int ldx, const float* dy, int ldy, int M, int K, int N, int accumulate, float* dw,
                         float* db, void* stream) {
  if (!x || !dy || !dw || !shapes_ok(ldx, ldy, M, K, N)) return HDRSKY_EINVAL;
  if (((uintptr_t)x | (uintptr_t)dy) & 15) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(fc_xtdy_kernel<false>, dim3(N / TN, K / TK), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, ldy, M, K,
                     N, dw, (float*)nullptr, 0.f, 0.f, 0.f, 1.f, (uint2*)nullptr, (unsigned short*)nullptr, db, accumulate);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_rmsprop_fc_fused(float* w, float* ms, const float* x, int ldx, const float* dy, int ldy, int M, int K, int N,
                            float lr, float rho, float eps, float gscale, void* packed_hi, void* natural_hi, float* db,
                            void* stream) {
  if (!w || !ms || !x || !dy || !packed_hi || !shapes_ok(ldx, ldy, M, K, N)) return HDRSKY_EINVAL;
  if (((uintptr_t)x | (uintptr_t)dy) & 15) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(fc_xtdy_kernel<true>, dim3(N / TN, K / TK), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, ldy, M, K,
                     N, w, ms, lr, rho, eps, gscale, (uint2*)packed_hi, (unsigned short*)natural_hi, db, 0);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

}  // extern "C"
