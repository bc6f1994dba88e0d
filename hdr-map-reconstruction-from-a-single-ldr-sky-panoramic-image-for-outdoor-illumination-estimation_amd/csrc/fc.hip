// Small-M (batch-sized) dense layers of the sun-pose net on the matrix cores: HBM-bound
// weight streaming.  out[s][m][o] = sum_{r in slice s} x[m][r] * Wop[r][o]
//   forward  (Keras Dense, sunpose_net.py:48-51,65-68):  r = input feature k, o = unit n,
//            weights read from the packed image [K/8][N][8] bf16 (8 consecutive k per 16 B)
//   dgrad    (tf.gradients through Dense, grad_cam.py:31): r = unit n, o = input feature k,
//            weights read from the natural image [K][N] bf16 (8 consecutive n per 16 B)
// Both are one kernel with two B-operand address strides.  Each workgroup owns 64 output
// columns x one reduction slice (split-R for occupancy: 64 column blocks x 4 slices = 256
// workgroups at N = 4096); weights go HBM -> VGPR directly (streamed once, not shared between
// waves), activations are staged as bf16 through a double-buffered LDS image shared by the 4 waves.
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "hooks.h"

namespace {

constexpr int RCH = 256;  // reduction elements per chunk (8 k-steps of 32)

// Finalisation inside the launch (hdrsky_fc_fwd_fin / hdrsky_fc_dgrad_fin): y[m][o] = act(bias[o] + sum_s part[s][m][o]), optional
// mask - what hdrsky_fc_finalize computes from the partial sums with a launch of its own (three of them sit on the dependent
// chain of the sun branch).  The workgroup that finishes LAST among the nsplit slices of a column block (a ticket per block)
// adds the slices in slice order - the same sum in the same order, whichever workgroup that is.  Visibility between workgroups
// on different dies (MI355X_MICROARCH.md, valid forms): the slices are stored write-through (`sc1`), drained (vmcnt(0)) ahead of
// the workgroup barrier and the ticket atomic; the last workgroup reads them with `sc1` loads - no fence, no L2 write-back.  The ticket resets itself for the next launch.
struct FcFin {
  float* y;                 // nullptr: partial sums only
  const float* bias;
  const float* mask_src;
  unsigned* tickets;        // [column blocks], zero before the first launch
  unsigned* zero_word;      // cleared by the launch (softmax_head's max accumulator further down the chain), or nullptr
  int relu;
};

// RG: reduction groups per workgroup.  A workgroup is RG x 4 waves: wave (rg, cw) owns 16 of the 64 output columns and the
// chunks ch = rg, rg + RG, ... of the workgroup's reduction slice; at the end the groups' accumulators are added through
// LDS in group order (fixed: bit-reproducible).  What bounds the kernel is the weight bytes in flight per CU (each wave
// keeps one 8 KB chunk in its registers and the next one in flight): at RG = 1 - one workgroup of four waves per CU - the
// stream ran at 2.8 TB/s; more reduction SLICES (hdrsky_fc_nsplit) raise it too, but every consumer of the partial sums then
// reads more slices.  RG = 4 (16 waves, 4 per SIMD, <= 128 VGPRs) has the bytes in flight of 16 slices and the partial sums of 4.
template <int MF, bool PRECISE, int RG>
__global__ void __launch_bounds__(256 * RG) fc_mfma_kernel(const float* __restrict__ x, const uint4* __restrict__ whi,
                                                           const uint4* __restrict__ wlo, float* __restrict__ out, int M,
                                                           int R, int O, int nsplit, long stride_col, long stride_kg, int nt, FcFin fin) {
  constexpr int MP = MF * 16, PL = PRECISE ? 2 : 1;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  // nt: the weight stream - every byte read once per launch by one workgroup - with the non-temporal policy: 100 MB per forward
  // (200 MB with the Grad-CAM sweep's data gradients) that otherwise evict the activations of the other branch from L2 / MALL.
  // Round 5 (profiles/r05_nt_more.txt): generator forward 0.510 -> 0.474 ms, training step -0.2 %.  HDRSKY_FC_W_NT=0: tuning hook.
  auto ldw = [&](const uint4* p) -> uint4 {
    if (nt) { const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    return *p;
  };
  extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
  uint4* sXall = reinterpret_cast<uint4*>(fc_smem);       // [RG][buf 2][hi/lo][kgroup(32)][m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rg = wave >> 2, cw = wave & 3, gtid = tid & 255;
  const int kq = lane >> 4, lr = lane & 15;
  const int ob = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int rslice = R / nsplit;
  const int r0 = sp * rslice;
  const int nchunks = (rslice + RCH - 1) / RCH;   // the last chunk may be partial (rslice % 8 == 0): zero-filled
  const int col = ob * 64 + cw * 16 + lr;
  const bool col_ok = col < O;
  const long wbase = (long)(col_ok ? col : 0) * stride_col;
  uint4* sX = sXall + (size_t)rg * 2 * PL * 32 * MP;

  f32x4_t acc[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto stage_x = [&](int chunk, int buf) {
    // items: MP rows x 32 k-groups of 8, by the 256 threads of this reduction group
    for (int i = gtid; i < MP * 32; i += 256) {
      const int m = i >> 5, kg = i & 31;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      if (m < M && chunk < nchunks && chunk * RCH + kg * 8 < rslice) {
        const float* p = x + (size_t)m * R + r0 + chunk * RCH + kg * 8;
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
      uint4 hi, lo;
      pack8<PRECISE>(v, hi, lo);
      sX[(buf * PL + 0) * 32 * MP + kg * MP + m] = hi;
      if (PRECISE) sX[(buf * PL + PL - 1) * 32 * MP + kg * MP + m] = lo;
    }
  };

  uint4 bh[8], bl[8];
  auto load_w = [&](int chunk) {
    const long kg0 = (long)(r0 + chunk * RCH) / 8;
    if ((chunk + 1) * RCH <= rslice) {      // (wave-uniform) full chunk: unguarded stream
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const long idx = wbase + (kg0 + s * 4 + kq) * stride_kg;
        bh[s] = ldw(whi + idx);
        if (PRECISE) bl[s] = ldw(wlo + idx);
      }
    } else {                                // partial last chunk (reduction lengths that are not multiples of 256), or none
      const int left = chunk < nchunks ? (rslice - chunk * RCH) >> 3 : 0;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bool ok = s * 4 + kq < left;
        const long idx = wbase + (ok ? kg0 + s * 4 + kq : 0) * stride_kg;
        // (loaded unconditionally from a clamped index, then masked: `ok ? whi[idx] : zero` compiled to a load through a
        // selected POINTER, with the zero constant parked in scratch memory)
        const uint4 th = whi[idx];
        bh[s] = make_uint4(ok ? th.x : 0u, ok ? th.y : 0u, ok ? th.z : 0u, ok ? th.w : 0u);
        if (PRECISE) {
          const uint4 tl = wlo[idx];
          bl[s] = make_uint4(ok ? tl.x : 0u, ok ? tl.y : 0u, ok ? tl.z : 0u, ok ? tl.w : 0u);
        }
      }
    }
  };

  // every group walks the same number of trips (barriers are workgroup-wide); a trip past the group's last chunk multiplies zeros
  const int ntrips = (nchunks + RG - 1) / RG;
  stage_x(rg, 0);
  load_w(rg);
  __syncthreads();
  for (int it = 0; it < ntrips; ++it) {
    const int buf = it & 1, nxt = (it + 1) * RG + rg;
    uint4 ch_bh[8], ch_bl[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { ch_bh[s] = bh[s]; if (PRECISE) ch_bl[s] = bl[s]; }
    if (it + 1 < ntrips) load_w(nxt);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const uint4 ah = sX[(buf * PL + 0) * 32 * MP + (s * 4 + kq) * MP + mf * 16 + lr];
        if (PRECISE) {
          const uint4 al = sX[(buf * PL + PL - 1) * 32 * MP + (s * 4 + kq) * MP + mf * 16 + lr];
          acc[mf] = mfma16(al, ch_bh[s], acc[mf]);
          acc[mf] = mfma16(ah, ch_bl[s], acc[mf]);
        }
        acc[mf] = mfma16(ah, ch_bh[s], acc[mf]);
      }
    }
    if (it + 1 < ntrips) stage_x(nxt, buf ^ 1);
    __syncthreads();
  }
  if (RG > 1) {
    // the groups' partial sums, added in group order by group 0 (the operand buffers are free: everyone is past the last barrier)
    f32x4_t* sAcc = reinterpret_cast<f32x4_t*>(fc_smem);          // [RG - 1][MF][256]
    if (rg > 0) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) sAcc[((rg - 1) * MF + mf) * 256 + gtid] = acc[mf];
    }
    __syncthreads();
    if (rg > 0 && fin.y == nullptr) return;
    if (rg == 0) {
#pragma unroll
      for (int g = 1; g < RG; ++g)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const f32x4_t t = sAcc[((g - 1) * MF + mf) * 256 + gtid];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[mf][j] += t[j];
        }
    }
  }
  if (col_ok && rg == 0) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = mf * 16 + kq * 4 + j;
        if (m < M) {
          float* dst = out + ((size_t)sp * M + m) * O + col;
          // (finalising launch: a write-through store - agent-scope relaxed atomic, `sc1` - so that the last workgroup's `sc1` loads
          // find the slice without an L2 write-back / invalidate: an agent-scope FENCE here flushes the whole L2 of the die and
          // cost the step 38 %, profiles/r05_fc_fin_ab.txt)
          if (fin.y != nullptr) __hip_atomic_store(dst, acc[mf][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *dst = acc[mf][j];
        }
      }
  }
  if (fin.y == nullptr) return;            // (kernel argument: uniform)
  // ---- finalisation by the last workgroup of this column block ---------------------------------------------------------
  __shared__ unsigned s_last;
  if (blockIdx.x == 0 && tid == 0 && fin.zero_word) *fin.zero_word = 0u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have landed ...
  __syncthreads();                                      // ... and every wave's of the workgroup, before its ticket
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(fin.tickets + ob, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned)(nsplit - 1)) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  for (int i = tid; i < M * 64; i += 256 * RG) {
    const int m = i >> 6, c = ob * 64 + (i & 63);
    if (c >= O) continue;
    float v = fin.bias ? fin.bias[c] : 0.f;
    for (int s2 = 0; s2 < nsplit; ++s2)
      v += __hip_atomic_load(out + ((size_t)s2 * M + m) * O + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fin.relu) v = fmaxf(v, 0.f);
    if (fin.mask_src) v = fin.mask_src[(size_t)m * O + c] > 0.f ? v : 0.f;
    fin.y[(size_t)m * O + c] = v;
  }
  if (tid == 0) __hip_atomic_store(fin.tickets + ob, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// fp32 [K][N] -> packed [K/8][N][8] bf16 (hi/lo) and natural [K][N] bf16 (hi/lo)
__global__ void fc_pack_kernel(const float* __restrict__ w, int K, int N, unsigned short* __restrict__ pk_hi,
                               unsigned short* __restrict__ pk_lo, unsigned short* __restrict__ nat_hi,
                               unsigned short* __restrict__ nat_lo) {
  const size_t total = (size_t)K * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = i % N;
    const size_t k = i / N;
    const float v = w[i];
    const unsigned short h = f2bf(v);
    const unsigned short l = f2bf(v - bf2f(h));
    const size_t pidx = ((k >> 3) * N + n) * 8 + (k & 7);
    if (pk_hi) pk_hi[pidx] = h;
    if (pk_lo) pk_lo[pidx] = l;
    if (nat_hi) nat_hi[i] = h;
    if (nat_lo) nat_lo[i] = l;
  }
}

template <int MF, bool PRECISE, int RG>
int launch_fc_v(const float* x, const void* whi, const void* wlo, float* out, int M, int R, int O, int nsplit,
                long stride_col, long stride_kg, hipStream_t s, const FcFin& fin) {
  const int grid = cdiv(O, 64) * nsplit;
  constexpr int lds = RG * 2 * (PRECISE ? 2 : 1) * 32 * (MF * 16) * 16;
  auto kern = fc_mfma_kernel<MF, PRECISE, RG>;
  if (lds > 64 * 1024) {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return HDRSKY_ELAUNCH;
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * RG), lds, s, x, (const uint4*)whi, (const uint4*)wlo, out, M, R, O, nsplit,
                     stride_col, stride_kg, hdrsky_hooks().fc_w_nt, fin);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

template <bool PRECISE>
int launch_fc(const float* x, const void* whi, const void* wlo, float* out, int M, int R, int O, int nsplit,
              long stride_col, long stride_kg, hipStream_t s, const FcFin& fin = FcFin{}) {
  // reduction groups per workgroup (tuning hook HDRSKY_FC_RG; the split-product mode holds twice the registers and LDS: 1)
  // Measured (profiles/r04_fc_rg_ab.txt, fc1 8192 -> 4096, 32 rows): forward 24.3 -> 16.7 us at 4 groups (2.8 -> 4.0 TB/s of
  // weight stream); the data gradient (natural image: 16-byte pieces of 4096 different rows per load) gains little beyond 2
  // (18.4 -> 17.1 us).  Forward pass 0.508 -> 0.504 ms; the training step does not notice.
  const int hook = hdrsky_hooks().fc_rg;
  const int rg = PRECISE ? 1 : (stride_col == 1 ? hook : (hook < 2 ? hook : 2));
  const int chunks = (R / nsplit + RCH - 1) / RCH;
  if (M <= 16) {
    if (rg >= 4 && chunks >= 4) return launch_fc_v<1, PRECISE, PRECISE ? 1 : 4>(x, whi, wlo, out, M, R, O, nsplit, stride_col, stride_kg, s, fin);
    if (rg >= 2 && chunks >= 2) return launch_fc_v<1, PRECISE, PRECISE ? 1 : 2>(x, whi, wlo, out, M, R, O, nsplit, stride_col, stride_kg, s, fin);
    return launch_fc_v<1, PRECISE, 1>(x, whi, wlo, out, M, R, O, nsplit, stride_col, stride_kg, s, fin);
  }
  if (rg >= 4 && chunks >= 4) return launch_fc_v<2, PRECISE, PRECISE ? 1 : 4>(x, whi, wlo, out, M, R, O, nsplit, stride_col, stride_kg, s, fin);
  if (rg >= 2 && chunks >= 2) return launch_fc_v<2, PRECISE, PRECISE ? 1 : 2>(x, whi, wlo, out, M, R, O, nsplit, stride_col, stride_kg, s, fin);
  return launch_fc_v<2, PRECISE, 1>(x, whi, wlo, out, M, R, O, nsplit, stride_col, stride_kg, s, fin);
}

}  // namespace

extern "C" {

int hdrsky_fc_pack_weights(const float* w, int K, int N, void* packed_hi, void* packed_lo, void* natural_hi,
                           void* natural_lo, void* stream) {
  if (!w || (K & 7) || (N & 7)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(fc_pack_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, w, K, N,
                     (unsigned short*)packed_hi, (unsigned short*)packed_lo, (unsigned short*)natural_hi,
                     (unsigned short*)natural_lo);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_fc_nsplit(int R) {
  // 64 column blocks x 4 slices = 256 workgroups at N = 4096.  8 slices (two workgroups per CU, 64 KB of weight loads in
  // flight per CU) stream the weights faster on their own (3.0 -> 4.1 TB/s), but every consumer of the partial sums
  // (fc_finalize, softmax_head, the next layer's staging) reads twice as many: inside the forward pass 4 is 12 us faster
  // and the training step does not care (profiles/ab_bench.sh; HDRSKY_FC_NSPLIT overrides for A/B runs).
  const int pref = hdrsky_hooks().fc_nsplit;   // (tuning hook; 4)
  int ns = pref > 0 ? pref : 4;
  while (ns > 1 && (R % (ns * RCH)) != 0) ns >>= 1;
  return ns;
}

// forward: out_part[nsplit][M][N] = x[M][K] @ W[K][N]   (bias / activation: hdrsky_fc_finalize or hdrsky_softmax_head)
int hdrsky_fc_fwd(const float* x, const void* packed_hi, const void* packed_lo, int M, int K, int N, int nsplit,
                  int compute, float* out_part, void* stream) {
  if (!x || !packed_hi || !out_part || M <= 0 || M > 32 || nsplit <= 0 || (K % (nsplit * 8)) != 0 || (N & 7))
    return HDRSKY_EINVAL;
  if (compute == HDRSKY_BF16X3) {
    if (!packed_lo) return HDRSKY_EINVAL;
    return launch_fc<true>(x, packed_hi, packed_lo, out_part, M, K, N, nsplit, 1, N, (hipStream_t)stream);
  }
  return launch_fc<false>(x, packed_hi, nullptr, out_part, M, K, N, nsplit, 1, N, (hipStream_t)stream);
}

// data gradient: dx_part[nsplit][M][K] = dy[M][N] @ W[K][N]^T   (weights in the natural bf16 image)
int hdrsky_fc_dgrad(const float* dy, const void* natural_hi, const void* natural_lo, int M, int K, int N, int nsplit,
                    int compute, float* dx_part, void* stream) {
  if (!dy || !natural_hi || !dx_part || M <= 0 || M > 32 || nsplit <= 0 || (N % (nsplit * 8)) != 0 || (K & 7))
    return HDRSKY_EINVAL;
  if (compute == HDRSKY_BF16X3) {
    if (!natural_lo) return HDRSKY_EINVAL;
    return launch_fc<true>(dy, natural_hi, natural_lo, dx_part, M, N, K, nsplit, N / 8, 1, (hipStream_t)stream);
  }
  return launch_fc<false>(dy, natural_hi, nullptr, dx_part, M, N, K, nsplit, N / 8, 1, (hipStream_t)stream);
}

// hdrsky_fc_fwd + hdrsky_fc_finalize in ONE launch: y[M][N] = act(bias + x @ W) (optional mask; zero_word cleared).  part_ws: the
// [nsplit][M][N] partial sums (workspace, also valid afterwards); tickets: cdiv(N, 64) zero-initialised words owned by this
// call site (they reset themselves; two launches that share them must not overlap).  Results bit-identical to the two launches.
int hdrsky_fc_fwd_fin(const float* x, const void* packed_hi, const void* packed_lo, int M, int K, int N, int nsplit, int compute,
                      float* part_ws, void* tickets, const float* bias, int relu, const float* mask_src, float* y, void* zero_word,
                      void* stream) {
  if (!x || !packed_hi || !part_ws || !tickets || !y || M <= 0 || M > 32 || nsplit <= 0 || (K % (nsplit * 8)) != 0 || (N & 7))
    return HDRSKY_EINVAL;
  const FcFin fin{y, bias, mask_src, (unsigned*)tickets, (unsigned*)zero_word, relu};
  if (compute == HDRSKY_BF16X3) {
    if (!packed_lo) return HDRSKY_EINVAL;
    return launch_fc<true>(x, packed_hi, packed_lo, part_ws, M, K, N, nsplit, 1, N, (hipStream_t)stream, fin);
  }
  return launch_fc<false>(x, packed_hi, nullptr, part_ws, M, K, N, nsplit, 1, N, (hipStream_t)stream, fin);
}

// the same for the data gradient: dx[M][K] = mask(dy @ W^T) (tickets: cdiv(K, 64) words)
int hdrsky_fc_dgrad_fin(const float* dy, const void* natural_hi, const void* natural_lo, int M, int K, int N, int nsplit, int compute,
                        float* part_ws, void* tickets, const float* bias, int relu, const float* mask_src, float* dx, void* zero_word,
                        void* stream) {
  if (!dy || !natural_hi || !part_ws || !tickets || !dx || M <= 0 || M > 32 || nsplit <= 0 || (N % (nsplit * 8)) != 0 || (K & 7))
    return HDRSKY_EINVAL;
  const FcFin fin{dx, bias, mask_src, (unsigned*)tickets, (unsigned*)zero_word, relu};
  if (compute == HDRSKY_BF16X3) {
    if (!natural_lo) return HDRSKY_EINVAL;
    return launch_fc<true>(dy, natural_hi, natural_lo, part_ws, M, N, K, nsplit, N / 8, 1, (hipStream_t)stream, fin);
  }
  return launch_fc<false>(dy, natural_hi, nullptr, part_ws, M, N, K, nsplit, N / 8, 1, (hipStream_t)stream, fin);
}

}  // extern "C"
