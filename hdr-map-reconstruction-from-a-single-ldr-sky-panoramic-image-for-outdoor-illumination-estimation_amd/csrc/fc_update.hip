// Dense weight gradient on the matrix cores, and the RMSprop update that never materialises it.
//
//   g[k][n] = sum_m x[m][k] * dy[m][n]          (tf.gradients through Keras Dense, sunpose_net.py:48-51,65-68;
//                                                optimizer: train.py:403 RMSprop(lr), rho 0.9, eps 1e-7)
//
// The two sun-pose Dense kernels hold 50.3 M of the 58.3 M trainables.  Written out, their gradient costs 168 MB of
// HBM writes and 168 MB of reads per step beside the 840 MB the update itself moves; the contraction behind it is
// 2*M flops per weight (M = batch rows, 32..256) - free next to that traffic.  So the update kernel recomputes the
// gradient tile it is about to apply.
//   launch 1 (fc_operands_kernel): x and dy -> bf16, transposed to [column][Mpad] so that the 8 consecutive rows an MFMA
//            lane needs are one 16-byte load; the bias gradient (fp32 column sums of dy) falls out of the same pass.
//   launch 2 (fc_xtdy_kernel): wave = 32 k x 64 n of the kernel, no LDS and no barriers - the operand fragments come
//            straight from L2 (1.5 MB for both matrices at M = 32), v_mfma_f32_32x32x16_bf16 contracts them, and every
//            lane then owns 32 weights whose w / ms it reads, updates and writes, together with the two bf16 MFMA images
//            of the new weights (packed [K/8][N][8] and natural [K][N], see fc.hip).  A workgroup is four such waves
//            side by side in n: 1 KB contiguous per kernel row.
// HBM-bound: 20 B per weight (w, ms read + written, 2 x 2 B images); algorithmic minimum of an RMSprop step.  Measured
// 4.9 TB/s on the 8192x4096 kernel: 8 B/weight of reads at ~6.5 TB/s plus 12 B/weight of writes at the ~4.4 TB/s a
// write-only stream reaches here (hdrsky_fc_wgrad_bf16's store epilogue) - the write side is what bounds it.
//
// The same contraction with a plain store epilogue (hdrsky_fc_wgrad_bf16) is the materialised gradient for callers that
// need it (a data-parallel all-reduce, gradient inspection).  Operands are rounded to bf16 (2^-9 relative per factor,
// fp32 accumulation) - the BF16 compute mode's contract, like every convolution weight gradient; the bias gradient is
// summed from the fp32 rows.  BF16X3 keeps hdrsky_fc_wgrad (fp32 FMA).
#include <cstdlib>

#include "common.h"
#include "hooks.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;


constexpr int WK = 32;              // kernel tile of one wave: 32 k x (NB 32x32 MFMA blocks side by side in n);
                                    // workgroup: four waves side by side in n
constexpr int MC = 32;              // batch rows are padded to a multiple of this

__device__ __forceinline__ f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0,
                                                 0, 0);
}

// src [M][ld] fp32 -> bf16 fragment image dst[C/32][Mpad/8][32 columns] of 16-byte groups (8 consecutive rows of one
// column; rows >= M zero): a wave's MFMA operand load is then two contiguous 512-byte runs.  blockIdx.z picks the matrix
// (0: x, 1: dy); thread = (column, group of 8 rows); colsum (dy only): fp32 sums over the M rows, in row order
struct OperandJob { const float* src; int ld, C; uint4* dst; float* colsum; int accumulate;
                    float* bw; float* bms; float lr, rho, eps, gscale; };   // bw != null: RMSprop of the bias vector with the column sums, here
__global__ void __launch_bounds__(256) fc_operands_kernel(OperandJob jx, OperandJob jd, int M, int Mpad) {
  const OperandJob j = blockIdx.z ? jd : jx;
  const int c = blockIdx.x * 256 + threadIdx.x, mg = blockIdx.y;
  if (c >= j.C) return;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = mg * 8 + i;
    v[i] = m < M ? j.src[(size_t)m * j.ld + c] : 0.f;
  }
  uint4 hi, lo;
  pack8<false>(v, hi, lo);
  j.dst[((size_t)(c >> 5) * (Mpad >> 3) + mg) * 32 + (c & 31)] = hi;
  if (j.colsum && mg == 0) {
    float s = 0.f;
    for (int m0 = 0; m0 < M; m0 += 8) {       // row order kept; eight rows' loads in flight (one per trip: M dependent round trips)
      float r8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r8[k] = j.src[(size_t)min(m0 + k, M - 1) * j.ld + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) if (m0 + k < M) s += r8[k];
    }
    j.colsum[c] = j.accumulate ? j.colsum[c] + s : s;
    if (j.bw != nullptr) {      // the Dense bias: its gradient is this column sum - updated here instead of by a launch of its own
      const float g = s * j.gscale;
      const float m_ = j.rho * j.bms[c] + (1.f - j.rho) * g * g;
      j.bms[c] = m_;
      j.bw[c] -= j.lr * g / (sqrtf(m_) + j.eps);
    }
  }
}

__device__ __forceinline__ unsigned swap_pairs(unsigned v) {   // lane 2j <-> lane 2j+1 (quad_perm [1,0,3,2])
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}

// NT (bit mask): non-temporal cache policy (aux = 2) on 1: the stores of the two bf16 images, 2: the stores of w / ms, 4: the loads
// of w / ms - bytes that are read or written exactly once per step and only evict what other launches would still find in L2 / MALL
template <bool FUSED, int NB, int NT = 0>
__global__ void __launch_bounds__(256, NB == 1 ? 8 : 5) fc_xtdy_kernel(const uint4* __restrict__ xT, const uint4* __restrict__ dT,
                                                         int Mpad, int K, int N, float* __restrict__ w,
                                                         float* __restrict__ ms, float lr, float rho, float eps,
                                                         float gscale, uint2* __restrict__ pk_hi,
                                                         unsigned* __restrict__ nat_hi, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  constexpr int AUX_IMG = (NT & 1) ? 2 : 0, AUX_ST = (NT & 2) ? 2 : 0, AUX_LD = (NT & 4) ? 2 : 0;
  constexpr int WN = 32 * NB;
  const int nw = (blockIdx.x * 4 + wave) * WN;
  const int mq = Mpad >> 3;           // 16-byte groups per operand row
  // gridDim.y < K / WK: a workgroup walks its k tiles (a launch capped to few workgroups that trickles along beside other
  // streams' work instead of taking the chip: hdrsky_hooks().fc_update_rows)
  for (int kb = blockIdx.y * WK; kb < K; kb += gridDim.y * WK) {

  f32x16_t acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

  const uint4* pa = xT + ((size_t)(kb >> 5) * mq + h) * 32 + r;
  const uint4* pb = dT + ((size_t)(nw >> 5) * mq + h) * 32 + r;
  for (int g = 0; g < mq; g += 2) {    // 16 rows of the batch per MFMA
    const uint4 a = pa[g * 32];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = mfma32(a, pb[((size_t)b * mq + g) * 32], acc[b]);
  }

  // D layout of 32x32: register i of lane (r, h) is row (i & 3) + 8 (i >> 2) + 4 h, column r.
  // Buffer addressing: the row offset is wave-uniform (SGPR soffset), the lane part one 32-bit voffset for all 16
  // registers of a block - 64-bit per-element addresses would cost 64 VGPRs here.
  const unsigned bytes = (unsigned)K * (unsigned)N * 4u;
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(w, 0, bytes, 0x00020000);
  const auto rm = __builtin_amdgcn_make_buffer_rsrc(ms, 0, FUSED ? bytes : 0u, 0x00020000);
  const auto rp = __builtin_amdgcn_make_buffer_rsrc(pk_hi, 0, FUSED ? bytes / 2 : 0u, 0x00020000);
  const auto rn = __builtin_amdgcn_make_buffer_rsrc(nat_hi, 0, (FUSED && nat_hi) ? bytes / 2 : 0u, 0x00020000);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int n = nw + b * 32 + r;
    const unsigned vo = ((unsigned)(4 * h) * (unsigned)N + (unsigned)n) * 4u;           // lane part, bytes
#define ROWB(i) ((unsigned)(kb + ((i) & 3) + 8 * ((i) >> 2)) * (unsigned)N * 4u)        // uniform part, bytes
    if (FUSED) {
      float wv[16], mv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        wv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vo, ROWB(i), AUX_LD));
        mv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, vo, ROWB(i), AUX_LD));
      }
      unsigned hb[16];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = half * 8; i < half * 8 + 8; ++i) {
          const float gg = acc[b][i] * gscale;
          const float m_ = rho * mv[i] + (1.f - rho) * gg * gg;
          const float nv = wv[i] - lr * gg / (sqrtf(m_) + eps);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m_), rm, vo, ROWB(i), AUX_ST);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nv), rw, vo, ROWB(i), AUX_ST);
          hb[i] = f2bf(nv);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // packed image [K/8][N][8] bf16: rows 8q + 4h .. +3 are half (8 bytes) of one 16-byte group
      const unsigned vp = (unsigned)n * 16u + (unsigned)h * 8u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
        const u32x2_t v = {hb[4 * q] | (hb[4 * q + 1] << 16), hb[4 * q + 2] | (hb[4 * q + 3] << 16)};
        __builtin_amdgcn_raw_buffer_store_b64(v, rp, vp, (unsigned)((kb >> 3) + q) * (unsigned)N * 16u, AUX_IMG);
      }
      // natural image [K][N] bf16: neighbouring lanes trade one value so that every lane stores a 32-bit pair
      // (even lane: row i, columns n, n+1; odd lane: row i+1, columns n-1, n); a NULL image has zero records (dropped)
      const bool odd = r & 1;
      const unsigned vn = ((unsigned)(4 * h + (odd ? 1 : 0)) * (unsigned)N + (unsigned)(n & ~1)) * 2u;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const unsigned got = swap_pairs(odd ? hb[i] : hb[i + 1]);
        const unsigned word = odd ? (got | (hb[i + 1] << 16)) : (hb[i] | (got << 16));
        __builtin_amdgcn_raw_buffer_store_b32(word, rn, vn, ROWB(i) >> 1, AUX_IMG);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = acc[b][i];
        if (accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vo, ROWB(i), 0));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rw, vo, ROWB(i), 0);
      }
    }
#undef ROWB
    __builtin_amdgcn_sched_barrier(0);   // one 32x32 block at a time: keeps the loads of the next out of this one's registers
  }
  }   // k tiles of this workgroup
}

inline bool shapes_ok(const float* x, int ldx, const float* dy, int ldy, int M, int K, int N) {
  return M > 0 && K > 0 && N > 0 && (size_t)K * N < ((size_t)1 << 30) && (K % WK) == 0 && (N % 256) == 0 && ldx >= K && ldy >= N && (ldx & 3) == 0 &&
         (ldy & 3) == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0;
}

inline int mpad(int M) { return roundup(M, MC); }

// 32x32 MFMA blocks per wave.  Two (82 VGPRs, 5 waves per SIMD, half the A-operand loads) measured 2 % faster at M = 32 and 13 %
// at M = 256 on the 8192x4096 kernel ALONE and was the default of rounds 3-4.  Inside the three-stream step ONE block per wave (58
// VGPRs, 8 waves per SIMD, twice the workgroups of half the footprint) is worth 1.1 % of the step (2.426 -> 2.398 ms,
// profiles/r05_fc_update_nb_ab.txt) and with the non-temporal image stores it is no slower alone either (114 us, 0.73 of the HBM
// peak): the default since round 5.  HDRSKY_FC_UPDATE_NB=2: tuning hook.
inline int blocks_per_wave() {
  return hdrsky_hooks().fc_update_nb == 2 ? 2 : 1;
}

template <bool FUSED>
inline void launch_xtdy(hipStream_t st, const uint4* xT, const uint4* dT, int Mp, int K, int N, float* w, float* ms, float lr,
                        float rho, float eps, float gscale, void* pk, void* nat, int accumulate) {
  const int nt = FUSED ? hdrsky_hooks().fc_nt : 0;
  const int cap = FUSED ? hdrsky_hooks().fc_update_rows : 0;      // (tuning hook HDRSKY_FC_UPDATE_ROWS: grid rows of the fused update; 0 = one per k tile)
  const int rows = (cap > 0 && cap < K / WK) ? cap : K / WK;
#define HDRSKY_XTDY_NT(NB_, NT_)                                                                                                      \
    case NT_: hipLaunchKernelGGL((fc_xtdy_kernel<FUSED, NB_, NT_>), dim3(N / (128 * NB_), rows), dim3(256), 0, st, xT, dT, Mp, K, N, w, \
                                 ms, lr, rho, eps, gscale, (uint2*)pk, (unsigned*)nat, accumulate); break;
  if (blocks_per_wave() == 2) {
    switch (nt) { HDRSKY_XTDY_NT(2, 1) HDRSKY_XTDY_NT(2, 2) HDRSKY_XTDY_NT(2, 3) HDRSKY_XTDY_NT(2, 4) HDRSKY_XTDY_NT(2, 5) HDRSKY_XTDY_NT(2, 6) HDRSKY_XTDY_NT(2, 7)
                  default: HDRSKY_XTDY_NT(2, 0) }
  } else {
    switch (nt) { HDRSKY_XTDY_NT(1, 1) default: HDRSKY_XTDY_NT(1, 0) }
  }
#undef HDRSKY_XTDY_NT
}

// launch 1 of both entry points; returns the transposed operand images inside ws
inline int operands(const float* x, int ldx, const float* dy, int ldy, int M, int K, int N, float* db, int accumulate,
                    void* ws, hipStream_t st, const uint4** xT, const uint4** dT, float* bw = nullptr, float* bms = nullptr,
                    float lr = 0.f, float rho = 0.f, float eps = 0.f, float gscale = 1.f) {
  const int Mp = mpad(M);
  uint4* px = (uint4*)ws;
  uint4* pd = px + (size_t)K * (Mp >> 3);
  OperandJob jx{x, ldx, K, px, nullptr, 0, nullptr, nullptr, 0.f, 0.f, 0.f, 1.f}, jd{dy, ldy, N, pd, db, accumulate, bw, bms, lr, rho, eps, gscale};
  hipLaunchKernelGGL(fc_operands_kernel, dim3(cdiv(K > N ? K : N, 256), Mp >> 3, 2), dim3(256), 0, st, jx, jd, M, Mp);
  HDRSKY_CHECK_LAUNCH();
  *xT = px; *dT = pd;
  return HDRSKY_OK;
}

}  // namespace

extern "C" {

size_t hdrsky_fc_xtdy_ws_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  return (size_t)(K + N) * mpad(M) * 2;
}

int hdrsky_fc_wgrad_bf16(const float* x, int ldx, const float* dy, int ldy, int M, int K, int N, int accumulate, float* dw,
                         float* db, void* ws, void* stream) {
  if (!x || !dy || !dw || !ws || !shapes_ok(x, ldx, dy, ldy, M, K, N)) return HDRSKY_EINVAL;
  const uint4 *xT, *dT;
  const int rc = operands(x, ldx, dy, ldy, M, K, N, db, accumulate, ws, (hipStream_t)stream, &xT, &dT);
  if (rc != HDRSKY_OK) return rc;
  launch_xtdy<false>((hipStream_t)stream, xT, dT, mpad(M), K, N, dw, nullptr, 0.f, 0.f, 0.f, 1.f, nullptr, nullptr, accumulate);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_rmsprop_fc_fused_bias(float* w, float* ms, const float* x, int ldx, const float* dy, int ldy, int M, int K, int N,
                                 float lr, float rho, float eps, float gscale, void* packed_hi, void* natural_hi, float* db,
                                 float* bias, float* bias_ms, void* ws, void* stream);

int hdrsky_rmsprop_fc_fused(float* w, float* ms, const float* x, int ldx, const float* dy, int ldy, int M, int K, int N,
                            float lr, float rho, float eps, float gscale, void* packed_hi, void* natural_hi, float* db,
                            void* ws, void* stream) {
  return hdrsky_rmsprop_fc_fused_bias(w, ms, x, ldx, dy, ldy, M, K, N, lr, rho, eps, gscale, packed_hi, natural_hi, db, nullptr, nullptr,
                                      ws, stream);
}

// ... with the layer's bias vector updated too (bias [N], bias_ms [N]; db [N] must be given: the column sums of dy are its gradient),
// by the operand launch that forms those sums - not by an hdrsky_rmsprop launch behind the update
int hdrsky_rmsprop_fc_fused_bias(float* w, float* ms, const float* x, int ldx, const float* dy, int ldy, int M, int K, int N,
                                 float lr, float rho, float eps, float gscale, void* packed_hi, void* natural_hi, float* db,
                                 float* bias, float* bias_ms, void* ws, void* stream) {
  if (!w || !ms || !x || !dy || !packed_hi || !ws || !shapes_ok(x, ldx, dy, ldy, M, K, N)) return HDRSKY_EINVAL;
  if ((bias != nullptr) != (bias_ms != nullptr) || (bias && !db)) return HDRSKY_EINVAL;
  const uint4 *xT, *dT;
  const int rc = operands(x, ldx, dy, ldy, M, K, N, db, 0, ws, (hipStream_t)stream, &xT, &dT, bias, bias_ms, lr, rho, eps, gscale);
  if (rc != HDRSKY_OK) return rc;
  launch_xtdy<true>((hipStream_t)stream, xT, dT, mpad(M), K, N, w, ms, lr, rho, eps, gscale, packed_hi, natural_hi, 0);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

// The fused update in two calls (round 5: the update itself deferred into the next step's forward pass, where one stream idles,
// instead of closing the step - trainer.Trainer(defer_dense=True)):
//   _prepare: launch 1 - x and dy to their transposed bf16 images inside ws, the bias gradient db (and, given bias / bias_ms, the
//             bias vector's RMSprop step).  After it the update depends on nothing but ws: x and dy may be rewritten.
//   _apply:   launch 2 - contracts the images of ws and updates w, ms and the two bf16 MFMA images.  The same launches with the same
//             arguments as hdrsky_rmsprop_fc_fused_bias issues back to back: bit-identical results.
int hdrsky_rmsprop_fc_fused_prepare(const float* x, int ldx, const float* dy, int ldy, int M, int K, int N, float lr, float rho, float eps,
                                    float gscale, float* db, float* bias, float* bias_ms, void* ws, void* stream) {
  if (!x || !dy || !ws || !shapes_ok(x, ldx, dy, ldy, M, K, N)) return HDRSKY_EINVAL;
  if ((bias != nullptr) != (bias_ms != nullptr) || (bias && !db)) return HDRSKY_EINVAL;
  const uint4 *xT, *dT;
  return operands(x, ldx, dy, ldy, M, K, N, db, 0, ws, (hipStream_t)stream, &xT, &dT, bias, bias_ms, lr, rho, eps, gscale);
}

int hdrsky_rmsprop_fc_fused_apply(float* w, float* ms, int M, int K, int N, float lr, float rho, float eps, float gscale, void* packed_hi,
                                  void* natural_hi, const void* ws, void* stream) {
  if (!w || !ms || !packed_hi || !ws || M <= 0 || K <= 0 || N <= 0 || (size_t)K * N >= ((size_t)1 << 30) || (K % WK) != 0 || (N % 256) != 0)
    return HDRSKY_EINVAL;
  const int Mp = mpad(M);
  const uint4* xT = (const uint4*)ws;
  const uint4* dT = xT + (size_t)K * (Mp >> 3);
  launch_xtdy<true>((hipStream_t)stream, xT, dT, Mp, K, N, w, ms, lr, rho, eps, gscale, packed_hi, natural_hi, 0);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

}  // extern "C"
