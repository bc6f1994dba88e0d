// JPEG quality round trip of the training augmentation (train.py:86-92: tf.image.adjust_jpeg_quality = libjpeg baseline
// encode, 4:2:0, slow-integer DCT, quality-scaled Annex-K tables, then decode with slow-integer IDCT and "fancy" chroma
// upsampling) without the lossless entropy-coding stage: pure integer / byte work, bit-exact against libjpeg.
//
// Two launches per batch:
//   jpeg_blocks_kernel   8 threads per 8x8 block (Y blocks at full resolution, Cb / Cr blocks of the 2x2-averaged
//                        planes): float -> u8 -> YCbCr (jccolor.c) -> [h2v2 downsample, jcsample.c] -> level shift ->
//                        jfdctint.c rows, LDS transpose, columns -> quantise (jcdctmgr.c) -> dequantise -> jidctint.c
//                        columns, LDS transpose, rows -> range limit -> one 8-byte store per thread into the u8 planes
//   jpeg_merge_kernel    one thread per four output pixels: h2v2 fancy (triangle) upsampling of the decoded chroma planes
//                        (jdsample.c) + YCbCr -> RGB (jdcolor.c) -> float / 255
// Any image size: partial 16x16 MCUs are completed by libjpeg's edge replication rules (see jpeg_blocks_kernel).
// Algorithmic bytes per image: 12 B/pixel read + 1.5 B/pixel plane write, then 1.5 (+ cached neighbours) read + 12 written.
#include <hip/hip_runtime.h>

#include "common.h"
#include "hdrsky.h"

namespace {

__constant__ unsigned char kStdLuma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                           14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                           18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                           49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
__constant__ unsigned char kStdChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                             24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                             99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633;
constexpr int F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c, one 8-point pass in place
template <bool FIRST>
__device__ __forceinline__ void fdct8(int (&d)[8]) {
  const int t0 = d[0] + d[7], t7 = d[0] - d[7], t1 = d[1] + d[6], t6 = d[1] - d[6];
  const int t2 = d[2] + d[5], t5 = d[2] - d[5], t3 = d[3] + d[4], t4 = d[3] - d[4];
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  constexpr int N = FIRST ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
  if (FIRST) { d[0] = (t10 + t11) << PASS1_BITS; d[4] = (t10 - t11) << PASS1_BITS; }
  else { d[0] = descale(t10 + t11, PASS1_BITS); d[4] = descale(t10 - t11, PASS1_BITS); }
  int z1 = (t12 + t13) * F_0_541;
  d[2] = descale(z1 + t13 * F_0_765, N);
  d[6] = descale(z1 - t12 * F_1_847, N);
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = (z3 + z4) * F_1_175;
  const int a4 = t4 * F_0_298, a5 = t5 * F_2_053, a6 = t6 * F_3_072, a7 = t7 * F_1_501;
  z1 = -z1 * F_0_899; z2 = -z2 * F_2_562; z3 = -z3 * F_1_961 + z5; z4 = -z4 * F_0_390 + z5;
  d[7] = descale(a4 + z1 + z3, N); d[5] = descale(a5 + z2 + z4, N);
  d[3] = descale(a6 + z2 + z3, N); d[1] = descale(a7 + z1 + z4, N);
}

// jidctint.c, one 8-point pass in place
template <bool FIRST>
__device__ __forceinline__ void idct8(int (&c)[8]) {
  int z1 = (c[2] + c[6]) * F_0_541;
  const int e2 = z1 - c[6] * F_1_847, e3 = z1 + c[2] * F_0_765;
  const int e0 = (c[0] + c[4]) << CONST_BITS, e1 = (c[0] - c[4]) << CONST_BITS;
  const int t10 = e0 + e3, t13 = e0 - e3, t11 = e1 + e2, t12 = e1 - e2;
  int t0 = c[7], t1 = c[5], t2 = c[3], t3 = c[1];
  z1 = t0 + t3;
  int z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
  const int z5 = (z3 + z4) * F_1_175;
  t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
  z1 = -z1 * F_0_899; z2 = -z2 * F_2_562; z3 = -z3 * F_1_961 + z5; z4 = -z4 * F_0_390 + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  constexpr int N = FIRST ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS + 3;
  c[0] = descale(t10 + t3, N); c[7] = descale(t10 - t3, N);
  c[1] = descale(t11 + t2, N); c[6] = descale(t11 - t2, N);
  c[2] = descale(t12 + t1, N); c[5] = descale(t12 - t1, N);
  c[3] = descale(t13 + t0, N); c[4] = descale(t13 - t0, N);
}

__device__ __forceinline__ int to_u8(float v) { return min(max((int)rintf(v * 255.f), 0), 255); }

// jccolor.c rgb_ycc_convert (16-bit fixed point).  comp: 0 Y, 1 Cb, 2 Cr
__device__ __forceinline__ int ycc(int r, int g, int b, int comp) {
  constexpr int HALF = 1 << 15, OFF = 128 << 16;
  if (comp == 0) return (19595 * r + 38470 * g + 7471 * b + HALF) >> 16;
  if (comp == 1) return (-11059 * r - 21709 * g + 32768 * b + OFF + HALF - 1) >> 16;
  return (32768 * r - 27439 * g - 5329 * b + OFF + HALF - 1) >> 16;
}

constexpr int BLK_PER_WG = 32;   // 256 threads

// Image sizes that are not multiples of 16 follow libjpeg's edge rules: full-resolution rows / columns are replicated
// (jcsample.c expand_right_edge, jcprepct.c expand_bottom_edge: rows only up to an even count), the DOWNSAMPLED chroma
// planes are then padded downwards by replicating their last row (jcprepct.c), the decoder crops.  The planes in `ws`
// have the padded sizes Hp x Wp (multiples of 16) and Hp/2 x Wp/2.
__global__ void __launch_bounds__(256) jpeg_blocks_kernel(const float* __restrict__ ldr, const int* __restrict__ quality,
                                                          int B, int H, int W, int Hp, int Wp, int bgr,
                                                          unsigned char* __restrict__ ws) {
  __shared__ int tile[BLK_PER_WG][8][9];
  const int lb = threadIdx.x >> 3, r = threadIdx.x & 7;
  const int ybl = (Hp >> 3) * (Wp >> 3), cbl = (Hp >> 4) * (Wp >> 4), per_img = ybl + 2 * cbl;
  const long long gb = (long long)blockIdx.x * BLK_PER_WG + lb;
  const bool live = gb < (long long)B * per_img;
  int d[8];
  int b = 0, comp = 0, by = 0, bx = 0;
  if (live) {
    b = (int)(gb / per_img);
    int k = (int)(gb % per_img);
    if (k >= ybl) { comp = 1 + (k - ybl) / cbl; k = (k - ybl) % cbl; by = k / (Wp >> 4); bx = k % (Wp >> 4); }
    else { by = k / (Wp >> 3); bx = k % (Wp >> 3); }
    const int ri = bgr ? 2 : 0, bi = bgr ? 0 : 2;
    const float* img = ldr + (size_t)b * H * W * 3;
    const bool vec = (W & 3) == 0;            // 16-byte aligned pixel rows: float4 loads for blocks inside the image
    if (comp == 0) {
      const int row = min(by * 8 + r, H - 1), col0 = bx * 8;
      float v[24];
      if (vec && col0 + 8 <= W) {
        const float4* src = reinterpret_cast<const float4*>(img + ((size_t)row * W + col0) * 3);
#pragma unroll
        for (int k = 0; k < 6; ++k) { const float4 t = src[k]; v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w; }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float* px = img + ((size_t)row * W + min(col0 + c, W - 1)) * 3;
          v[c * 3] = px[0]; v[c * 3 + 1] = px[1]; v[c * 3 + 2] = px[2];
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) d[c] = ycc(to_u8(v[c * 3 + ri]), to_u8(v[c * 3 + 1]), to_u8(v[c * 3 + bi]), 0) - 128;
    } else {   // jcsample.c h2v2_downsample: (a + b + c + d + bias) >> 2, bias 1,2,1,2,... along the row
      const int hc = (H + 1) >> 1;
      const int crow = min(by * 8 + r, hc - 1);                    // rows below the downsampled image replicate its last row
      const int row0 = 2 * crow, row1 = min(2 * crow + 1, H - 1), col0 = bx * 16;
      int acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = 0;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = rr ? row1 : row0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {      // 8 source pixels (24 floats) at a time
          float v[24];
          if (vec && col0 + h * 8 + 8 <= W) {
            const float4* sp = reinterpret_cast<const float4*>(img + ((size_t)row * W + col0 + h * 8) * 3);
#pragma unroll
            for (int k = 0; k < 6; ++k) { const float4 t = sp[k]; v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float* px = img + ((size_t)row * W + min(col0 + h * 8 + e, W - 1)) * 3;
              v[e * 3] = px[0]; v[e * 3 + 1] = px[1]; v[e * 3 + 2] = px[2];
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[h * 4 + e / 2] += ycc(to_u8(v[e * 3 + ri]), to_u8(v[e * 3 + 1]), to_u8(v[e * 3 + bi]), comp);
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) d[c] = ((acc[c] + 1 + (c & 1)) >> 2) - 128;
    }
    fdct8<true>(d);                      // row r
#pragma unroll
    for (int c = 0; c < 8; ++c) tile[lb][r][c] = d[c];
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = tile[lb][k][r];    // column r
    fdct8<false>(d);
    // jcparam.c jpeg_quality_scaling + force_baseline; jcdctmgr.c: divisor q << 3, round half away from zero
    const int q = min(max(quality[b], 1), 100);
    const int scale = q < 50 ? 5000 / q : 200 - 2 * q;
    const unsigned char* std_tbl = comp == 0 ? kStdLuma : kStdChroma;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int qv = min(max((std_tbl[k * 8 + r] * scale + 50) / 100, 1), 255);
      const int div = qv << 3, a = abs(d[k]);
      const int m = (a + (div >> 1)) / div;
      d[k] = (d[k] < 0 ? -m : m) * qv;    // quantised, then dequantised (jidctint.c DEQUANTIZE)
    }
    idct8<true>(d);                      // column r
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[lb][k][r] = d[k];
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int c = 0; c < 8; ++c) d[c] = tile[lb][r][c];    // row r
    idct8<false>(d);
    unsigned int lo = 0, hi = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      lo |= (unsigned)min(max(d[c] + 128, 0), 255) << (8 * c);
      hi |= (unsigned)min(max(d[c + 4] + 128, 0), 255) << (8 * c);
    }
    const size_t ypl = (size_t)Hp * Wp, cpl = ypl >> 2;
    unsigned char* plane = ws + (size_t)b * (ypl + 2 * cpl) + (comp == 0 ? 0 : ypl + (comp - 1) * cpl);
    const int pw = comp == 0 ? Wp : (Wp >> 1);
    *reinterpret_cast<uint2*>(plane + (size_t)(by * 8 + r) * pw + bx * 8) = uint2{lo, hi};
  }
}

// jdsample.c for output pixels x0..x0+3 of row y of a decoded half-resolution plane (hc x wc samples, row stride cs):
// h2v2_fancy_upsample (triangle filter), or plain 2x2 replication when the plane has at most two columns
// (jinit_upsampler selects the fancy method only for downsampled_width > 2).
__device__ __forceinline__ void chroma4(const unsigned char* __restrict__ p, int hc, int wc, int cs, int y, int x0, int (&o)[4]) {
  const int cy = y >> 1, c0 = x0 >> 1, c1 = min(c0 + 1, wc - 1);
  const unsigned char* ra = p + (size_t)cy * cs;
  if (wc <= 2) { o[0] = o[1] = ra[c0]; o[2] = o[3] = ra[c1]; return; }
  const int oy = (y & 1) ? min(cy + 1, hc - 1) : max(cy - 1, 0);     // the nearer neighbour row (edge: replicated)
  const int cm = max(c0 - 1, 0), cp = min(c0 + 2, wc - 1);
  const unsigned char* rb = p + (size_t)oy * cs;
  const int sm = 3 * ra[cm] + rb[cm], s0 = 3 * ra[c0] + rb[c0], s1 = 3 * ra[c1] + rb[c1], sp = 3 * ra[cp] + rb[cp];
  const int s1n = (c0 + 1 <= wc - 1) ? s1 : s0;   // (only read for x0+2, x0+3, which do not exist then)
  o[0] = (3 * s0 + sm + 8) >> 4;
  o[1] = (3 * s0 + s1n + 7) >> 4;
  o[2] = (3 * s1 + s0 + 8) >> 4;
  o[3] = (3 * s1 + sp + 7) >> 4;
}

// One thread per group of four pixels of a row (the last group of a row may be shorter).
__global__ void __launch_bounds__(256) jpeg_merge_kernel(const unsigned char* __restrict__ ws, int B, int H, int W, int Hp,
                                                         int Wp, int bgr, float* __restrict__ out) {
  const size_t ypl = (size_t)Hp * Wp, cpl = ypl >> 2;
  const int gpr = (W + 3) >> 2;                            // groups per row
  const size_t total = (size_t)B * H * gpr;
  const int hc = (H + 1) >> 1, wc = (W + 1) >> 1;
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / ((size_t)H * gpr)), rem = (int)(g % ((size_t)H * gpr)), y = rem / gpr, x0 = (rem % gpr) * 4;
    const int n = min(4, W - x0);
    const unsigned char* base = ws + (size_t)b * (ypl + 2 * cpl);
    const unsigned int y4 = *reinterpret_cast<const unsigned int*>(base + (size_t)y * Wp + x0);   // Wp % 16 == 0: aligned, in bounds
    int cb[4], cr[4];
    chroma4(base + ypl, hc, wc, Wp >> 1, y, x0, cb);
    chroma4(base + ypl + cpl, hc, wc, Wp >> 1, y, x0, cr);
    float v[12];
    constexpr int HALF = 1 << 15;   // jdcolor.c ycc_rgb_convert
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int yy = (int)((y4 >> (8 * e)) & 255u), u = cb[e] - 128, w = cr[e] - 128;
      const int rr = min(max(yy + ((91881 * w + HALF) >> 16), 0), 255);
      const int gg = min(max(yy + ((-22554 * u - 46802 * w + HALF) >> 16), 0), 255);
      const int bb = min(max(yy + ((116130 * u + HALF) >> 16), 0), 255);
      v[e * 3 + (bgr ? 2 : 0)] = (float)rr / 255.f;
      v[e * 3 + 1] = (float)gg / 255.f;
      v[e * 3 + (bgr ? 0 : 2)] = (float)bb / 255.f;
    }
    float* o = out + (((size_t)b * H + y) * W + x0) * 3;
    if (n == 4 && (W & 3) == 0) {
      float4* o4 = reinterpret_cast<float4*>(o);
      o4[0] = float4{v[0], v[1], v[2], v[3]};
      o4[1] = float4{v[4], v[5], v[6], v[7]};
      o4[2] = float4{v[8], v[9], v[10], v[11]};
    } else {
      for (int e = 0; e < n * 3; ++e) o[e] = v[e];
    }
  }
}

}  // namespace

extern "C" {

size_t hdrsky_jpeg_roundtrip_ws_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const size_t Hp = ((size_t)H + 15) / 16 * 16, Wp = ((size_t)W + 15) / 16 * 16;
  return (size_t)B * Hp * Wp * 3 / 2;
}

int hdrsky_jpeg_roundtrip(const float* ldr, const int* quality, int B, int H, int W, int bgr, unsigned char* ws, float* out,
                          void* stream) {
  if (!ldr || !quality || !ws || !out || B <= 0 || H <= 0 || W <= 0) return HDRSKY_EINVAL;
  if ((reinterpret_cast<uintptr_t>(ws) & 7) != 0 || (reinterpret_cast<uintptr_t>(ldr) & 15) != 0 ||
      (reinterpret_cast<uintptr_t>(out) & 15) != 0)
    return HDRSKY_EINVAL;
  const int Hp = (H + 15) / 16 * 16, Wp = (W + 15) / 16 * 16;
  const long long nblocks = (long long)B * ((Hp >> 3) * (Wp >> 3) + 2 * (Hp >> 4) * (Wp >> 4));
  const long long g1 = (nblocks + BLK_PER_WG - 1) / BLK_PER_WG;
  if (g1 > 0x7fffffffLL) return HDRSKY_EUNSUPPORTED;
  hipLaunchKernelGGL(jpeg_blocks_kernel, dim3((unsigned)g1), dim3(256), 0, (hipStream_t)stream, ldr, quality, B, H, W, Hp, Wp,
                     bgr ? 1 : 0, ws);
  HDRSKY_CHECK_LAUNCH();
  size_t g2 = ((size_t)B * H * ((W + 3) / 4) + 255) / 256;
  if (g2 > 8192) g2 = 8192;
  hipLaunchKernelGGL(jpeg_merge_kernel, dim3((unsigned)g2), dim3(256), 0, (hipStream_t)stream, ws, B, H, W, Hp, Wp, bgr ? 1 : 0,
                     out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

}  // extern "C"
