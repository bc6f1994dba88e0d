// Minimal kernel for the a15 defect: does a dependent chain of packed-f32 VALU instructions (v_pk_fma_f32, what
// hipcc's SLP vectoriser makes of the distortion-aware gather's blend) deliver wrong values when other waves on the SAME
// SIMD issue MFMAs?  Victim waves run an exact counting chain A <- 1 * 1 + A with the dependent instructions D issue slots
// apart (D = 2 is what the compiler emits: one independent instruction in between, its 1-wait-state rule for VOP3P
// results); aggressor waves of the same workgroup (wave w + 4k shares the SIMD of wave w) run back-to-back
// v_mfma_f32_16x16x32_bf16.  Every lane's final count must equal the number of steps; the host lists the lanes that differ.
//   hipcc --offload-arch=gfx950 -O2 pk_hazard.hip -o pk_hazard && ./pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A = v[10:11], B = v[12:13], C = v[18:19]; X = v[14:15], W = v[16:17] (all ones)
#define FA "v_pk_fma_f32 v[10:11], v[14:15], v[16:17], v[10:11]\n"
#define FB "v_pk_fma_f32 v[12:13], v[14:15], v[16:17], v[12:13]\n"
#define FC "v_pk_fma_f32 v[18:19], v[14:15], v[16:17], v[18:19]\n"
// the compiler's form: a scalar weight broadcast from the low half of W
#define GA "v_pk_fma_f32 v[10:11], v[14:15], v[16:17], v[10:11] op_sel_hi:[1,0,1]\n"
#define GB "v_pk_fma_f32 v[12:13], v[14:15], v[16:17], v[12:13] op_sel_hi:[1,0,1]\n"
// plain VALU control
#define SA "v_fma_f32 v10, v14, v16, v10\n v_fma_f32 v11, v15, v17, v11\n"
#define SB "v_fma_f32 v12, v14, v16, v12\n v_fma_f32 v13, v15, v17, v13\n"
// the blend's tail: two chains summed by v_pk_add_f32 and converted (D = 2 everywhere), then re-seeded
#define R4(x) x x x x
#define R16(x) R4(R4(x))

#define VICTIM_FN(NAME, BODY)                                                                                       \
  __device__ __forceinline__ void NAME(int iters, float one, float& ax, float& ay, float& bx, float& by) {          \
    asm volatile(                                                                                                   \
        "v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n" \
        "v_mov_b32 v14, %4\n v_mov_b32 v15, %4\n v_mov_b32 v16, %4\n v_mov_b32 v17, %4\n"                          \
        "s_mov_b32 s20, %5\n"                                                                                       \
        "s_nop 4\n"                                                                                                 \
        "1:\n" BODY                                                                                                 \
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                         \
        "s_nop 7\n s_nop 7\n"                                                                                       \
        "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n v_mov_b32 %2, v12\n v_mov_b32 %3, v13\n"                          \
        : "=v"(ax), "=v"(ay), "=v"(bx), "=v"(by)                                                                    \
        : "v"(one), "s"(iters)                                                                                      \
        : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "s20", "scc", "memory");           \
  }

VICTIM_FN(victim_d1, R16(FA))                      // 16 steps of A per body, back to back
VICTIM_FN(victim_d2, R16(FA FB))                   // A and B alternate: 16 steps each, dependent pairs 2 slots apart
VICTIM_FN(victim_d3, R16(FA FB FC))                // three chains: 3 slots apart
VICTIM_FN(victim_d2_opsel, R16(GA GB))             // D = 2 with the broadcast modifier
VICTIM_FN(victim_d2_nop, R16(FA "s_nop 0\n" FB "s_nop 0\n"))   // D = 2 + one wait state more
VICTIM_FN(victim_scalar, R16(SA SB))               // plain v_fma_f32 control

// VALU writes VCC, the very next instruction is a SALU read of VCC (what the gather's `si >= 0` tests compile to:
// v_cmp_lt_i32 vcc, -1, v1 ; s_and_b64 s[56:57], s[4:5], vcc): are all 64 bits current?  The compare result alternates
// between all ones and all zeros, so a stale bit shows up in the error masks (returned as two floats' bit patterns).
__device__ __forceinline__ void victim_vcc(int iters, float one, float& ax, float& ay, float& bx, float& by) {
  unsigned elo, ehi;
  asm volatile(
      "s_mov_b64 s[26:27], 0\n s_mov_b32 s20, %2\n v_mov_b32 v10, %3\n s_nop 4\n"
      "1:\n"
      R16("v_cmp_eq_u32 vcc, v10, v10\n s_andn2_b64 s[22:23], exec, vcc\n s_or_b64 s[26:27], s[26:27], s[22:23]\n"
          "v_cmp_ne_u32 vcc, v10, v10\n s_and_b64 s[24:25], exec, vcc\n s_or_b64 s[26:27], s[26:27], s[24:25]\n")
      "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
      "v_mov_b32 %0, s26\n v_mov_b32 %1, s27\n"
      : "=v"(elo), "=v"(ehi)
      : "s"(iters), "v"(one)
      : "v10", "s20", "s22", "s23", "s24", "s25", "s26", "s27", "vcc", "scc", "memory");
  // expected value of every output is 16 * iters: encode "no error" as that, an error as the mask's bit count
  const float ok = 16.0f * iters;
  ax = elo == 0 ? ok : (float)__builtin_popcount(elo); ay = ehi == 0 ? ok : -(float)__builtin_popcount(ehi);
  bx = ok; by = ok;
}

// The instruction the forensics of the a15 defect point at (profiles/experiments/da_dbg/run_da_dbg5.py): v_pk_mul_f32 with
// op_sel:[0,1] - the LOW result takes the HIGH dword of src1 - delivered 0 in its low half in lanes 48-63.  Here: lo =
// v14 * v17, hi = v15 * v17 with v14 = 3, v15 = 5, v16 = 7, v17 = 11, checked against 33 / 55 after every pair; the error
// lane masks are OR-ed into s[26:27] (low-half errors) and s[28:29] (high-half errors).  The second form mimics the
// compiler's stream (two interleaved chains seeded by a cross-half multiply, D = 2).
#define XCHK "v_pk_mul_f32 v[10:11], v[14:15], v[16:17] op_sel:[0,1]\n v_pk_mul_f32 v[12:13], v[14:15], v[16:17] op_sel:[0,1]\n" \
             "v_cmp_neq_f32 vcc, v10, v18\n s_or_b64 s[26:27], s[26:27], vcc\n v_cmp_neq_f32 vcc, v11, v19\n s_or_b64 s[28:29], s[28:29], vcc\n" \
             "v_cmp_neq_f32 vcc, v12, v18\n s_or_b64 s[26:27], s[26:27], vcc\n v_cmp_neq_f32 vcc, v13, v19\n s_or_b64 s[28:29], s[28:29], vcc\n"
#define XMIM "v_mov_b32 v30, v17\n" \
             "v_pk_mul_f32 v[10:11], v[14:15], v[16:17] op_sel:[0,1]\n v_pk_mul_f32 v[12:13], v[14:15], v[30:31] op_sel_hi:[1,0]\n" \
             "v_pk_fma_f32 v[10:11], v[14:15], v[16:17], v[10:11] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[12:13], v[14:15], v[16:17], v[12:13] op_sel_hi:[1,0,1]\n" \
             "v_pk_fma_f32 v[10:11], v[14:15], v[16:17], v[10:11] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[12:13], v[14:15], v[16:17], v[12:13] op_sel_hi:[1,0,1]\n" \
             "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n s_nop 0\n" \
             "v_cmp_neq_f32 vcc, v10, v32\n s_or_b64 s[26:27], s[26:27], vcc\n v_cmp_neq_f32 vcc, v11, v33\n s_or_b64 s[28:29], s[28:29], vcc\n"
#define XVICTIM(NAME, BODY)                                                                                         \
  __device__ __forceinline__ void NAME(int iters, float one, float& ax, float& ay, float& bx, float& by) {          \
    unsigned llo, lhi, hlo, hhi;                                                                                    \
    asm volatile(                                                                                                   \
        "s_mov_b64 s[26:27], 0\n s_mov_b64 s[28:29], 0\n s_mov_b32 s20, %4\n"                                       \
        "v_mov_b32 v14, 0x40400000\n v_mov_b32 v15, 0x40a00000\n v_mov_b32 v16, 0x40e00000\n v_mov_b32 v17, 0x41300000\n" \
        "v_mov_b32 v18, 0x42040000\n v_mov_b32 v19, 0x425c0000\n v_mov_b32 v31, 0\n"                                \
        /* mimic: lo = 3*11 + 3*7 + 3*7 + (3*11 + 3*7 + 3*7) = 150 ; hi = 5*11 + 5*7 + 5*7 + (5*11 + 5*7 + 5*7) = 250 */ \
        "v_mov_b32 v32, 0x43160000\n v_mov_b32 v33, 0x437a0000\n s_nop 4\n"                                         \
        "1:\n" BODY                                                                                                 \
        "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                         \
        "v_mov_b32 %0, s26\n v_mov_b32 %1, s27\n v_mov_b32 %2, s28\n v_mov_b32 %3, s29\n"                          \
        : "=v"(llo), "=v"(lhi), "=v"(hlo), "=v"(hhi)                                                                \
        : "s"(iters)                                                                                                \
        : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v30", "v31", "v32", "v33", "s20", "s26", "s27",  \
          "s28", "s29", "vcc", "scc", "memory");                                                                    \
    const float ok = 16.0f * iters;   /* "no error" = the value the host expects; an error = (lane bit count) */       \
    ax = llo == 0 ? ok : 1000.f + (float)__builtin_popcount(llo); ay = lhi == 0 ? ok : 2000.f + (float)__builtin_popcount(lhi >> 16) * 100.f + (float)__builtin_popcount(lhi & 0xffff); \
    bx = hlo == 0 ? ok : 3000.f + (float)__builtin_popcount(hlo); by = hhi == 0 ? ok : 4000.f + (float)__builtin_popcount(hhi);     \
  }
XVICTIM(victim_xsel, R16(XCHK))
XVICTIM(victim_xmim, R16(XMIM))

// MFMA aggressor whose A operand comes from LDS every iteration (ds_read_b128 -> MFMA, as an implicit-GEMM conv wave)
__device__ __forceinline__ void aggressor_lds_mfma(int iters) {
  asm volatile(
      "v_mov_b32 v20, 0x3f803f80\n v_mov_b32 v21, 0x3f803f80\n v_mov_b32 v22, 0x3f803f80\n v_mov_b32 v23, 0x3f803f80\n"
      "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
      "v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n"
      "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n"
      "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"
      "v_mbcnt_lo_u32_b32 v40, -1, 0\n v_mbcnt_hi_u32_b32 v40, -1, v40\n v_lshlrev_b32 v40, 4, v40\n"
      "s_mov_b32 s21, %0\n s_nop 4\n"
      "2:\n"
      "ds_read_b128 v[42:45], v40\n ds_read_b128 v[46:49], v40 offset:1024\n ds_read_b128 v[50:53], v40 offset:2048\n ds_read_b128 v[54:57], v40 offset:3072\n"
      "s_waitcnt lgkmcnt(3)\n v_mfma_f32_16x16x32_bf16 v[24:27], v[42:45], v[20:23], v[24:27]\n"
      "s_waitcnt lgkmcnt(2)\n v_mfma_f32_16x16x32_bf16 v[28:31], v[46:49], v[20:23], v[28:31]\n"
      "s_waitcnt lgkmcnt(1)\n v_mfma_f32_16x16x32_bf16 v[32:35], v[50:53], v[20:23], v[32:35]\n"
      "s_waitcnt lgkmcnt(0)\n v_mfma_f32_16x16x32_bf16 v[36:39], v[54:57], v[20:23], v[36:39]\n"
      "v_mfma_f32_16x16x32_bf16 v[24:27], v[42:45], v[20:23], v[24:27]\n"
      "v_mfma_f32_16x16x32_bf16 v[28:31], v[46:49], v[20:23], v[28:31]\n"
      "v_mfma_f32_16x16x32_bf16 v[32:35], v[50:53], v[20:23], v[32:35]\n"
      "v_mfma_f32_16x16x32_bf16 v[36:39], v[54:57], v[20:23], v[36:39]\n"
      "s_sub_u32 s21, s21, 1\n s_cmp_lg_u32 s21, 0\n s_cbranch_scc1 2b\n"
      "s_nop 7\n s_nop 7\n s_nop 7\n"
      :
      : "s"(iters)
      : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36",
        "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
        "v54", "v55", "v56", "v57", "s21", "scc", "memory");
}

__device__ __forceinline__ void aggressor_mfma(int iters) {
  asm volatile(
      "v_mov_b32 v20, 0x3f803f80\n v_mov_b32 v21, 0x3f803f80\n v_mov_b32 v22, 0x3f803f80\n v_mov_b32 v23, 0x3f803f80\n"
      "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
      "v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n"
      "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n"
      "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"
      "s_mov_b32 s21, %0\n s_nop 4\n"
      "2:\n"
      "v_mfma_f32_16x16x32_bf16 v[24:27], v[20:23], v[20:23], v[24:27]\n"
      "v_mfma_f32_16x16x32_bf16 v[28:31], v[20:23], v[20:23], v[28:31]\n"
      "v_mfma_f32_16x16x32_bf16 v[32:35], v[20:23], v[20:23], v[32:35]\n"
      "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[20:23], v[36:39]\n"
      "v_mfma_f32_16x16x32_bf16 v[24:27], v[20:23], v[20:23], v[24:27]\n"
      "v_mfma_f32_16x16x32_bf16 v[28:31], v[20:23], v[20:23], v[28:31]\n"
      "v_mfma_f32_16x16x32_bf16 v[32:35], v[20:23], v[20:23], v[32:35]\n"
      "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[20:23], v[36:39]\n"
      "s_sub_u32 s21, s21, 1\n s_cmp_lg_u32 s21, 0\n s_cbranch_scc1 2b\n"
      "s_nop 7\n s_nop 7\n s_nop 7\n"
      :
      : "s"(iters)
      : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36",
        "v37", "v38", "v39", "s21", "scc", "memory");
}

__device__ __forceinline__ void aggressor_valu(int iters) {
  asm volatile(
      "v_mov_b32 v20, 1.0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
      "s_mov_b32 s21, %0\n s_nop 4\n"
      "2:\n"
      "v_fma_f32 v24, v20, v20, v24\n v_fma_f32 v25, v20, v20, v25\n v_fma_f32 v26, v20, v20, v26\n v_fma_f32 v27, v20, v20, v27\n"
      "v_fma_f32 v24, v20, v20, v24\n v_fma_f32 v25, v20, v20, v25\n v_fma_f32 v26, v20, v20, v26\n v_fma_f32 v27, v20, v20, v27\n"
      "s_sub_u32 s21, s21, 1\n s_cmp_lg_u32 s21, 0\n s_cbranch_scc1 2b\n"
      :
      : "s"(iters)
      : "v20", "v24", "v25", "v26", "v27", "s21", "scc", "memory");
}

// block = 64 * (4 + 4 * naggr) threads: waves 0-3 victims, the rest aggressors (kind: 1 MFMA, 2 plain VALU)
template <int PATTERN>
__global__ void __launch_bounds__(1024) hazard_kernel(float* out, int iters, int aggr_iters, int kind, float one) {
  __shared__ float4 s_lds[1024];
  s_lds[threadIdx.x & 1023] = float4{1.f, 1.f, 1.f, 1.f};
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    float ax, ay, bx, by;
    if (PATTERN == 1) victim_d1(iters, one, ax, ay, bx, by);
    else if (PATTERN == 2) victim_d2(iters, one, ax, ay, bx, by);
    else if (PATTERN == 3) victim_d3(iters, one, ax, ay, bx, by);
    else if (PATTERN == 4) victim_d2_opsel(iters, one, ax, ay, bx, by);
    else if (PATTERN == 5) victim_d2_nop(iters, one, ax, ay, bx, by);
    else if (PATTERN == 7) victim_vcc(iters, one, ax, ay, bx, by);
    else if (PATTERN == 8) victim_xsel(iters, one, ax, ay, bx, by);
    else if (PATTERN == 9) victim_xmim(iters, one, ax, ay, bx, by);
    else victim_scalar(iters, one, ax, ay, bx, by);
    float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    o[0] = ax; o[1] = ay; o[2] = bx; o[3] = by;
  } else {
    if (kind == 1) aggressor_mfma(aggr_iters);
    else if (kind == 3) aggressor_lds_mfma(aggr_iters);
    else aggressor_valu(aggr_iters);
  }
}

template <int PATTERN>
static void run(const char* name, int naggr, int kind, int iters, int nblocks, float* d_out, std::vector<float>& h) {
  const int threads = 64 * (4 + 4 * naggr);
  CHECK(hipMemset(d_out, 0, (size_t)nblocks * 256 * 4 * sizeof(float)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(hazard_kernel<PATTERN>, dim3(nblocks), dim3(threads), 0, 0, d_out, iters, iters * 2, kind, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(h.data(), d_out, (size_t)nblocks * 256 * 4 * sizeof(float), hipMemcpyDeviceToHost));
  const float expect_a = 16.0f * iters;
  size_t bad = 0; std::set<int> lanes; float worst = 0;
  const bool has_b = PATTERN != 1;
  for (size_t t = 0; t < (size_t)nblocks * 256; ++t)
    for (int c = 0; c < (has_b ? 4 : 2); ++c) {
      const float v = h[t * 4 + c];
      if (v != expect_a) { ++bad; lanes.insert((int)(t & 63)); if (expect_a - v > worst) worst = expect_a - v; }
    }
  printf("%-16s aggressors/SIMD %d (%s): %8.2f ms  wrong values %zu of %zu; steps lost at most %.0f; lanes:", name, naggr,
         naggr == 0 ? "none" : (kind == 1 ? "mfma" : (kind == 3 ? "lds+mfma" : "valu")), ms, bad, (size_t)nblocks * 256 * (has_b ? 4 : 2), worst);
  int shown = 0;
  for (int l : lanes) { if (shown++ < 70) printf(" %d", l); }
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int nblocks = argc > 2 ? atoi(argv[2]) : 1024;
  float* d_out; CHECK(hipMalloc(&d_out, (size_t)nblocks * 256 * 4 * sizeof(float)));
  std::vector<float> h((size_t)nblocks * 256 * 4);
  printf("steps per chain %d, workgroups %d\n", iters * 16, nblocks);
  for (int naggr = 0; naggr <= 2; ++naggr)
    for (int kind = 1; kind <= (naggr ? 3 : 1); ++kind) {
      run<6>("scalar D=2", naggr, kind, iters, nblocks, d_out, h);
      run<1>("pk D=1", naggr, kind, iters, nblocks, d_out, h);
      run<2>("pk D=2", naggr, kind, iters, nblocks, d_out, h);
      run<4>("pk D=2 op_sel", naggr, kind, iters, nblocks, d_out, h);
      run<5>("pk D=2 + nop", naggr, kind, iters, nblocks, d_out, h);
      run<3>("pk D=3", naggr, kind, iters, nblocks, d_out, h);
      run<7>("vcc->salu", naggr, kind, iters, nblocks, d_out, h);
      run<8>("pk_mul op_sel:[0,1]", naggr, kind, iters, nblocks, d_out, h);
      run<9>("compiler stream", naggr, kind, iters, nblocks, d_out, h);
    }
  return 0;
}
