// The matmul of a distortion-aware layer on its written gathered operand (distortion_aware_ops.py:107-121:
// tf.matmul(gathered [B*H*W, k*k*C], kernel [k*k*C, F]) + bias), i.e. a 1x1 convolution over K = k*k*C channels of a final
// bf16 NHWC tensor:   y[m][n] = sum_k A[m][k] * W[k][n] + bias[n],   m = (sample, pixel).
//
// Why not conv_igemm_kernel: its operand planes hold one power-of-two channel group at a time (1152 = 9 x 128: nine sequential
// stage -> barrier -> 4 k-steps -> barrier rounds per tile, every round exposing a round trip to memory: 61 us for the 9.7 GFLOP
// of a 128-channel layer on 32x128 maps at batch 8).  Here the operand is final bf16 - nothing to convert, normalise or pad - so
// both operands go global -> LDS by LDS-DMA into a ring of G1_NBUF stages of 64 k, three stages (96 KB per CU) in flight under the
// MFMAs of the current one, one barrier per stage:
//   stage = A tile 128 pixels x 64 k (16 KB, 16-byte slots XOR-swizzled by the pixel's low bits: the 16 lanes of an A-fragment
//           read hit distinct bank groups although a pixel's row is 128 B) + B tile 64 k x BN filters straight from the packed
//           filter image [k-step][4][Npad][8] (the k x k filter's own image: its k-steps are (tap, 32-channel block) in order);
//   4 waves = 2 x 2, a wave owns 64 pixels x BN/2 filters (4 x NI accumulator fragments), D[pixel][filter];
//   epilogue through an LDS tile (aliases the ring): + bias, InstanceNorm partials [B][HW/128][2][N] of the fp32 values in the
//   layout hdrsky_conv2d_fwd emits, fp32 or bf16 rows of 16-byte pieces.
// Bound: the A stream (each element read once: 75 MB per 128-channel layer above) and LDS reads (16 KB per wave and stage for 32
// MFMAs = 128 B/clk per CU).
#include <atomic>

#include "common.h"

namespace {

constexpr int G1_BM = 128, G1_BK = 64, G1_NBUF = 4;

struct G1Args {
  const unsigned short* A;   // [M][K] bf16
  const uint4* whi;          // packed filter image, hi plane
  const float* bias;         // [N] or null
  float* y;                  // [M][N] fp32 (or bf16 when y_bf16)
  float* stats;              // [B][HW/128][2][N] or null
  int y_bf16;
  int M, K, N, Npad, HW, nst, nblk;
};

__device__ __forceinline__ void g1_wait_vmcnt(int n) {   // n: wave-uniform
  switch (n) {
#define HDRSKY_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    HDRSKY_VMC(0) HDRSKY_VMC(2) HDRSKY_VMC(4) HDRSKY_VMC(5) HDRSKY_VMC(6) HDRSKY_VMC(8) HDRSKY_VMC(10) HDRSKY_VMC(12) HDRSKY_VMC(16)
#undef HDRSKY_VMC
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int NI>   // BN = 32 * NI filters per workgroup
__global__ void __launch_bounds__(256) gemm1x1_kernel(const G1Args a) {
  constexpr int BN = 32 * NI, BNP = BN + 4;
  constexpr int A_SLOTS = G1_BM * 8, B_SLOTS = 2 * 4 * BN;       // 16-byte slots of a stage
  constexpr int STAGE = (A_SLOTS + B_SLOTS) * 16;
  constexpr int APT = A_SLOTS / 256, BPT = B_SLOTS / 256, PER = APT + BPT;   // DMA instructions per thread and stage
  static_assert(B_SLOTS % 256 == 0 && G1_NBUF == 4, "stage geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, kq = lane >> 4;
  const int nt = blockIdx.x % a.nblk, mt = blockIdx.x / a.nblk;
  const int m0 = mt * G1_BM, n0 = nt * BN;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // DMA sources of this thread's slots (slot = j * 256 + tid), advanced by the stage
  const unsigned short* asrc[APT];
  const uint4* bsrc[BPT];
#pragma unroll
  for (int j = 0; j < APT; ++j) {
    const int slot = j * 256 + tid, p = slot >> 3, c = (slot & 7) ^ (p & 7);
    asrc[j] = a.A + (size_t)(m0 + p) * a.K + c * 8;
  }
#pragma unroll
  for (int j = 0; j < BPT; ++j) {
    const int slot = j * 256 + tid, r = slot / BN, n = slot % BN;      // r = (k-step of the stage) * 4 + q
    bsrc[j] = a.whi + (size_t)r * a.Npad + n0 + n;
  }
  auto issue = [&](int st) {
    const unsigned base = lds0 + (unsigned)(st & (G1_NBUF - 1)) * STAGE + (unsigned)wave * 1024u;
#pragma unroll
    for (int j = 0; j < APT; ++j) glds16(asrc[j] + (size_t)st * G1_BK, base + j * 4096);
#pragma unroll
    for (int j = 0; j < BPT; ++j) glds16(bsrc[j] + (size_t)st * 8 * a.Npad, base + A_SLOTS * 16 + j * 4096);
  };

  f32x4_t acc[4][NI];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment offsets inside a stage: A slot of (pixel m, 16-byte chunk c) = m * 8 + (c ^ (m & 7)); B slot = (k-step * 4 + kq) * BN + n
  int aoff[4][2], boff[NI][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) aoff[mi][kk] = ((wm * 64 + mi * 16 + lr) * 8 + ((kk * 4 + kq) ^ (lr & 7))) * 16;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) boff[ni][kk] = (A_SLOTS + (kk * 4 + kq) * BN + wn * (16 * NI) + ni * 16 + lr) * 16;

  const int nst = a.nst;
  for (int p = 0; p < G1_NBUF - 1 && p < nst; ++p) issue(p);
  for (int s = 0; s < nst; ++s) {
    const int ahead = min(G1_NBUF - 2, nst - 1 - s);      // stages in flight behind stage s
    g1_wait_vmcnt(ahead * PER);
    __builtin_amdgcn_s_barrier();                          // stage s is in LDS for everybody; everybody is past stage s - 1
    if (s + G1_NBUF - 1 < nst) issue(s + G1_NBUF - 1);     // ... whose ring slot is refilled
    const unsigned char* sb = smem + (s & (G1_NBUF - 1)) * STAGE;
    uint4 ah[2][4], bh[2][NI];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) ah[kk][mi] = *reinterpret_cast<const uint4*>(sb + aoff[mi][kk]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bh[kk][ni] = *reinterpret_cast<const uint4*>(sb + boff[ni][kk]);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma16(ah[kk][mi], bh[kk][ni], acc[mi][ni]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // every wave is done reading the ring: the epilogue tile aliases it

  // ---- epilogue: accumulators -> LDS tile [128][BN + 4] -> rows of 16-byte pieces -------------------------------------------------
  float* sOut = reinterpret_cast<float*>(smem);
  float* sStat = reinterpret_cast<float*>(smem + ((G1_BM * BNP * 4 + 15) & ~15));     // [wave 4][BN][2]
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        sOut[(wm * 64 + mi * 16 + kq * 4 + j) * BNP + wn * (16 * NI) + ni * 16 + lr] = acc[mi][ni][j];
  __syncthreads();
  constexpr int C4 = BN / 4, PPI = 256 / C4;
  const int c4 = tid % C4, n = n0 + c4 * 4;
  float bias4[4], cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) bias4[e] = a.bias != nullptr ? a.bias[n + e] : 0.f;
#pragma unroll
  for (int it = 0; it < G1_BM / PPI; ++it) {
    const int m = it * PPI + tid / C4;
    const float4 t = *reinterpret_cast<const float4*>(sOut + m * BNP + c4 * 4);
    const float v[4] = {t.x + bias4[0], t.y + bias4[1], t.z + bias4[2], t.w + bias4[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) { cs[e] += v[e]; cq[e] += v[e] * v[e]; }
    const size_t idx = (size_t)(m0 + m) * a.N + n;
    if (a.y_bf16)
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.y) + idx) =
          uint2{(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
    else
      *reinterpret_cast<float4*>(a.y + idx) = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (a.stats != nullptr) {
    // threads holding the same float4 column are C4 apart inside a wave
#pragma unroll
    for (int o = C4; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { cs[e] += __shfl_xor(cs[e], o); cq[e] += __shfl_xor(cq[e], o); }
    }
    if (lane < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sStat[((tid >> 6) * BN + c4 * 4 + e) * 2 + 0] = cs[e];
        sStat[((tid >> 6) * BN + c4 * 4 + e) * 2 + 1] = cq[e];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s += sStat[(w * BN + tid) * 2 + 0]; q += sStat[(w * BN + tid) * 2 + 1]; }
      const int b = m0 / a.HW, tile = (m0 - b * a.HW) / G1_BM, nparts = a.HW / G1_BM;
      float* dst = a.stats + ((size_t)(b * nparts + tile) * 2) * a.N + n0 + tid;
      dst[0] = s;
      dst[a.N] = q;
    }
  }
}

template <int NI>
int g1_launch(const G1Args& a, hipStream_t st) {
  constexpr int BN = 32 * NI;
  constexpr int lds = G1_NBUF * (G1_BM * 8 + 2 * 4 * BN) * 16;
  static_assert(lds >= G1_BM * (BN + 4) * 4 + 4 * BN * 2 * 4 + 16, "the epilogue tile aliases the ring");
  auto kern = gemm1x1_kernel<NI>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return HDRSKY_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.M / G1_BM) * a.nblk), dim3(256), lds, st, a);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

}  // namespace

extern "C" {

// [host] 1 when hdrsky_gemm1x1_bf16 takes the shape: whole 128-pixel tiles per sample, K a multiple of 64, N of 32
int hdrsky_gemm1x1_supported(int HW, int K, int N) {
  return HW > 0 && (HW % G1_BM) == 0 && K >= G1_BK && (K % G1_BK) == 0 && N >= 32 && (N % 32) == 0;
}

// [host] InstanceNorm partial rows per sample that hdrsky_gemm1x1_bf16 writes
int hdrsky_gemm1x1_stats_nparts(int HW) { return HW / G1_BM; }

int hdrsky_gemm1x1_bf16(const void* A, const void* w_hi, const float* bias, int B, int HW, int K, int N, void* y, int y_bf16,
                        float* stats_part, void* stream) {
  if (!A || !w_hi || !y || B <= 0) return HDRSKY_EINVAL;
  if (!hdrsky_gemm1x1_supported(HW, K, N)) return HDRSKY_EUNSUPPORTED;
  if ((size_t)B * HW >= ((size_t)1 << 31)) return HDRSKY_EUNSUPPORTED;
  G1Args a{};
  a.A = (const unsigned short*)A; a.whi = (const uint4*)w_hi; a.bias = bias; a.y = (float*)y; a.stats = stats_part; a.y_bf16 = y_bf16;
  a.M = B * HW; a.K = K; a.N = N; a.Npad = roundup(N, 64); a.HW = HW; a.nst = K / G1_BK;
  hipStream_t st = (hipStream_t)stream;
  if ((N % 128) == 0) { a.nblk = N / 128; return g1_launch<4>(a, st); }
  if ((N % 64) == 0) { a.nblk = N / 64; return g1_launch<2>(a, st); }
  a.nblk = N / 32;
  return g1_launch<1>(a, st);
}

}  // extern "C"
