// Implicit-GEMM NHWC convolution on gfx950 matrix cores (v_mfma_f32_16x16x32_bf16).
//
// One workgroup (4 waves) owns an output tile of TH x TW pixels of one sample and BN output
// channels.  The input halo patch of that tile is staged ONCE into LDS as bf16 channel-chunk
// planes ([Cin/8][halo pixel] of 16-byte units) with the producer's normalisation + activation
// (and, for the resize-deconv, the 2x bilinear resize; for stride-2 dgrad, the zero stuffing)
// applied on the way in.  The im2col matrix is never materialised: an MFMA A fragment for tap
// (ky,kx) is a ds_read_b128 of the halo plane at a shifted pixel address.  Weights are
// pre-packed into the B-fragment image [kstep][4][Npad][8] bf16 and streamed through a
// double-buffered LDS ring, KC k-steps per barrier.
//
// Replaces tf.nn.conv2d + bias_add (ops.py:41-42), resize+conv (ops.py:121-124), Keras Conv2D
// (discriminator.py:11-13, sunrad_net.py:12-14), vgg16.conv2d (vgg16.py:32-36) and, through
// transposed/flipped packed filters, their data gradients.
#include "common.h"

namespace {

constexpr int KC = 4;  // k-steps (of 32) per B chunk / barrier

struct ConvKArgs {
  const float* x;
  const uint4* whi;
  const uint4* wlo;
  const float* bias;
  const float* in_scale;
  const float* in_shift;
  const float* in_part;
  const float* in_gamma;
  const float* in_beta;
  const float* residual;
  float* y;
  float* stats;
  int B, H, W, Cin, Ho, Wo, Cout, Npad;
  int KH, KW, stride, pad_t, pad_l, upsample, dilate, Hc, Wc;
  int in_mode, ss_bstride, in_nparts;
  float in_eps, in_inv_count, in_slope, out_slope;
  int final_relu, want_stats;
  int tiles_x, tiles_y, nblocks;
  int cgs, log2nq, ngroups, ksg, log2cbg, ntaps;
  int HT, WT, NPIX, NPIXP, wt_magic;
  int off_alo, off_b, off_ss, off_tap, off_stat;
};

// load 8 consecutive channels and apply the producer's affine + leaky activation
__device__ __forceinline__ void load8_xf(const float* __restrict__ p, const float* sc, const float* sh, bool xf,
                                         float slope, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  if (xf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = leaky(v[j] * sc[j] + sh[j], slope);
  } else if (slope != 1.f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = leaky(v[j], slope);
  }
}

template <int WM, int WN, int MI, int NI, int TW, bool NARROW, bool PRECISE>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvKArgs a) {
  constexpr int BM = WM * MI * 16;
  constexpr int BN = WN * NI * 16;
  constexpr int TH = BM / TW;
  constexpr int FPR = TW / 16;               // M fragments per tile row
  constexpr int BITEMS = KC * 4 * BN;        // uint4 per B chunk plane
  constexpr int BPT = BITEMS / 256;          // per thread
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BITEMS % 256 == 0, "B chunk must tile the block");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* sAhi = reinterpret_cast<uint4*>(smem);
  uint4* sAlo = reinterpret_cast<uint4*>(smem + a.off_alo);
  uint4* sB = reinterpret_cast<uint4*>(smem + a.off_b);  // [2 bufs][hi,lo][BITEMS]
  float* sScale = reinterpret_cast<float*>(smem + a.off_ss);
  float* sShift = sScale + a.Cin;
  int* sTap = reinterpret_cast<int*>(smem + a.off_tap);
  float* sStat = reinterpret_cast<float*>(smem + a.off_stat);
  constexpr int BPLANES = PRECISE ? 2 : 1;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kq = lane >> 4, lr = lane & 15;

  int bid = blockIdx.x;
  const int nb = bid % a.nblocks; bid /= a.nblocks;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = nb * BN;
  const int iy0 = oy0 * a.stride - a.pad_t, ix0 = ox0 * a.stride - a.pad_l;

  // ---- prologue: tap offset table + input transform tables ---------------------------------
  for (int t = tid; t < a.ntaps; t += 256) sTap[t] = (t / a.KW) * a.WT + (t % a.KW);
  const bool xf = a.in_mode != HDRSKY_IN_NONE;
  if (a.in_mode == HDRSKY_IN_AFFINE) {
    for (int c = tid; c < a.Cin; c += 256) {
      sScale[c] = a.in_scale[b * a.ss_bstride + c];
      sShift[c] = a.in_shift[b * a.ss_bstride + c];
    }
  } else if (a.in_mode == HDRSKY_IN_PARTIALS) {
    for (int c = tid; c < a.Cin; c += 256) {
      float s = 0.f, ss = 0.f;
      const float* pp = a.in_part + (size_t)b * a.in_nparts * 2 * a.Cin + c;
      for (int p = 0; p < a.in_nparts; ++p) {
        s += pp[(2 * p) * a.Cin];
        ss += pp[(2 * p + 1) * a.Cin];
      }
      const float mean = s * a.in_inv_count;
      const float var = fmaxf(ss * a.in_inv_count - mean * mean, 0.f);
      const float inv = a.in_gamma[c] / sqrtf(var + a.in_eps);
      sScale[c] = inv;
      sShift[c] = a.in_beta[c] - mean * inv;
    }
  }
  __syncthreads();

  // ---- per-lane fragment bases ---------------------------------------------------------------
  int abase[MI];  // byte offset of this lane's A row (pixel) inside a plane (+ its k-quarter plane)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int f = wm * MI + mi;
    const int fr = f / FPR, fc = f % FPR;
    const int pix = (fr * a.stride) * a.WT + (fc * 16 + lr) * a.stride;
    abase[mi] = (NARROW ? pix : (kq * a.NPIXP + pix)) * 16;
  }
  int bbase[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bbase[ni] = (kq * BN + (wn * NI + ni) * 16 + lr) * 16;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nchunks = (a.ksg + KC - 1) / KC;
  const int cin32 = a.Cin >> 5;

  for (int g = 0; g < a.ngroups; ++g) {
    if (g > 0) __syncthreads();  // everyone finished reading the previous group's planes

    // ---- stage the halo patch of this channel group into LDS --------------------------------
    if (NARROW) {
      for (int p = tid; p < a.NPIX; p += 256) {
        const int hy = (int)(((unsigned)p * (unsigned)a.wt_magic) >> 24);
        const int hx = p - hy * a.WT;
        const int cy = iy0 + hy, cx = ix0 + hx;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc) {
          const float* src = a.x + ((size_t)(b * a.H + cy) * a.W + cx) * a.Cin;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < a.Cin) {
              float t = src[j];
              if (xf) t = t * sScale[j] + sShift[j];
              v[j] = leaky(t, a.in_slope);
            }
        }
        uint4 hi, lo;
        pack8<PRECISE>(v, hi, lo);
        sAhi[p] = hi;
        if (PRECISE) sAlo[p] = lo;
      }
    } else {
      const int nq = 1 << a.log2nq;
      const int nitems = a.NPIX << a.log2nq;
      for (int i = tid; i < nitems; i += 256) {
        const int p = i >> a.log2nq, q = i & (nq - 1);
        const int hy = (int)(((unsigned)p * (unsigned)a.wt_magic) >> 24);
        const int hx = p - hy * a.WT;
        const int cy = iy0 + hy, cx = ix0 + hx;
        const int c0 = g * a.cgs + q * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc) {
          const float* sc = sScale + c0;
          const float* sh = sShift + c0;
          const float* xb = a.x + (size_t)b * a.H * a.W * a.Cin + c0;
          if (a.upsample == 2) {
            // tf.image.resize BILINEAR, half-pixel centres: src = (dst+0.5)*0.5-0.5
            const float sy = (cy + 0.5f) * 0.5f - 0.5f, sx = (cx + 0.5f) * 0.5f - 0.5f;
            const float fy = floorf(sy), fx = floorf(sx);
            const int ylo = max((int)fy, 0), yhi = min((int)ceilf(sy), a.H - 1);
            const int xlo = max((int)fx, 0), xhi = min((int)ceilf(sx), a.W - 1);
            const float ly = sy - fy, lx = sx - fx;
            float tl[8], tr[8], bl[8], br[8];
            load8_xf(xb + ((size_t)ylo * a.W + xlo) * a.Cin, sc, sh, xf, a.in_slope, tl);
            load8_xf(xb + ((size_t)ylo * a.W + xhi) * a.Cin, sc, sh, xf, a.in_slope, tr);
            load8_xf(xb + ((size_t)yhi * a.W + xlo) * a.Cin, sc, sh, xf, a.in_slope, bl);
            load8_xf(xb + ((size_t)yhi * a.W + xhi) * a.Cin, sc, sh, xf, a.in_slope, br);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float top = tl[j] + (tr[j] - tl[j]) * lx;
              const float bot = bl[j] + (br[j] - bl[j]) * lx;
              v[j] = top + (bot - top) * ly;
            }
          } else if (a.dilate == 2) {
            if (((cy | cx) & 1) == 0)
              load8_xf(xb + ((size_t)(cy >> 1) * a.W + (cx >> 1)) * a.Cin, sc, sh, xf, a.in_slope, v);
          } else {
            load8_xf(xb + ((size_t)cy * a.W + cx) * a.Cin, sc, sh, xf, a.in_slope, v);
          }
        }
        uint4 hi, lo;
        pack8<PRECISE>(v, hi, lo);
        sAhi[q * a.NPIXP + p] = hi;
        if (PRECISE) sAlo[q * a.NPIXP + p] = lo;
      }
    }

    // ---- B ring: chunk 0 ------------------------------------------------------------------------
    uint4 breg[BPLANES][BPT];
    auto load_b = [&](int chunk) {
#pragma unroll
      for (int j = 0; j < BPT; ++j) {
        const int i = tid + j * 256;
        const int ksl = i / (4 * BN);
        const int rem = i - ksl * (4 * BN);
        const int q = rem / BN, n = rem - q * BN;
        const int ks = chunk * KC + ksl;
        uint4 vh = uint4{0, 0, 0, 0}, vl = uint4{0, 0, 0, 0};
        if (ks < a.ksg) {
          int kp;
          if (NARROW) kp = ks;
          else kp = (ks >> a.log2cbg) * cin32 + (g << a.log2cbg) + (ks & ((1 << a.log2cbg) - 1));
          const size_t src = ((size_t)(kp * 4 + q) * a.Npad + n0 + n);
          vh = a.whi[src];
          if (PRECISE) vl = a.wlo[src];
        }
        breg[0][j] = vh;
        if (PRECISE) breg[BPLANES - 1][j] = vl;
      }
    };
    auto store_b = [&](int buf) {
#pragma unroll
      for (int j = 0; j < BPT; ++j) {
        sB[(buf * BPLANES + 0) * BITEMS + tid + j * 256] = breg[0][j];
        if (PRECISE) sB[(buf * BPLANES + 1) * BITEMS + tid + j * 256] = breg[BPLANES - 1][j];
      }
    };
    load_b(0);
    store_b(0);
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
      const int buf = ch & 1;
      if (ch + 1 < nchunks) load_b(ch + 1);
      const unsigned char* bh = reinterpret_cast<const unsigned char*>(sB + (buf * BPLANES) * BITEMS);
      const unsigned char* bl = reinterpret_cast<const unsigned char*>(sB + (buf * BPLANES + 1) * BITEMS);
#pragma unroll
      for (int ksl = 0; ksl < KC; ++ksl) {
        const int ks = ch * KC + ksl;
        if (ks < a.ksg) {
          int aoff;
          if (NARROW) {
            const int tap = min(ks * 4 + kq, a.ntaps - 1);
            aoff = sTap[tap] * 16;
          } else {
            const int tap = ks >> a.log2cbg, cb = ks & ((1 << a.log2cbg) - 1);
            aoff = (sTap[tap] + cb * 4 * a.NPIXP) * 16;
          }
          uint4 ah[MI], al[MI], wh[NI], wl[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            ah[mi] = *reinterpret_cast<const uint4*>(smem + abase[mi] + aoff);
            if (PRECISE) al[mi] = *reinterpret_cast<const uint4*>(smem + a.off_alo + abase[mi] + aoff);
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            wh[ni] = *reinterpret_cast<const uint4*>(bh + bbase[ni] + ksl * (4 * BN * 16));
            if (PRECISE) wl[ni] = *reinterpret_cast<const uint4*>(bl + bbase[ni] + ksl * (4 * BN * 16));
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              if (PRECISE) {
                acc[mi][ni] = mfma16(al[mi], wh[ni], acc[mi][ni]);
                acc[mi][ni] = mfma16(ah[mi], wl[ni], acc[mi][ni]);
              }
              acc[mi][ni] = mfma16(ah[mi], wh[ni], acc[mi][ni]);
            }
        }
      }
      if (ch + 1 < nchunks) store_b(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------
  float csum[NI], csq[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) { csum[ni] = 0.f; csq[ni] = 0.f; }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + (wn * NI + ni) * 16 + lr;
    const bool nok = n < a.Cout;
    const float bv = (a.bias != nullptr && nok) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int f = wm * MI + mi;
      const int oy = oy0 + f / FPR;
      const int oxb = ox0 + (f % FPR) * 16 + kq * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = oxb + j;
        if (nok && oy < a.Ho && ox < a.Wo) {
          float v = acc[mi][ni][j] + bv;
          csum[ni] += v;
          csq[ni] += v * v;
          const size_t idx = ((size_t)(b * a.Ho + oy) * a.Wo + ox) * a.Cout + n;
          if (a.out_slope != 1.f) v = leaky(v, a.out_slope);
          if (a.residual != nullptr) v += a.residual[idx];
          if (a.final_relu) v = fmaxf(v, 0.f);
          a.y[idx] = v;
        }
      }
    }
  }
  if (a.want_stats) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float s = csum[ni], q = csq[ni];
      s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
      if (kq == 0) {
        const int col = (wn * NI + ni) * 16 + lr;
        sStat[(wm * BN + col) * 2 + 0] = s;
        sStat[(wm * BN + col) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s += sStat[(w * BN + tid) * 2 + 0];
        q += sStat[(w * BN + tid) * 2 + 1];
      }
      const int nparts = a.tiles_x * a.tiles_y;
      float* dst = a.stats + ((size_t)(b * nparts + ty * a.tiles_x + tx) * 2) * a.Cout + n0 + tid;
      dst[0] = s;
      dst[a.Cout] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// weight packing: fp32 HWIO -> [kstep][4][Npad][8] bf16 (hi / lo planes)
// ---------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, int KH, int KW, int Cin, int Cout, int Npad,
                                    int narrow, int flip, int ksteps, unsigned short* __restrict__ hi,
                                    unsigned short* __restrict__ lo) {
  const size_t total = (size_t)ksteps * 4 * Npad * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7;
    size_t r = i >> 3;
    const int n = r % Npad; r /= Npad;
    const int q = r & 3;
    const int kp = (int)(r >> 2);
    int tap, c;
    if (narrow) { tap = kp * 4 + q; c = j; }
    else { const int cin32 = Cin >> 5; tap = kp / cin32; c = (kp % cin32) * 32 + q * 8 + j; }
    float v = 0.f;
    if (tap < KH * KW && c < Cin && n < Cout) {
      int ky = tap / KW, kx = tap % KW;
      if (flip) {
        // packed filter w'[ky,kx,c(=co of w),n(=ci of w)] = w[KH-1-ky, KW-1-kx, n, c]; w is [KH,KW,Cout',Cin'] = [.., n-range, c-range]
        ky = KH - 1 - ky; kx = KW - 1 - kx;
        v = w[((size_t)(ky * KW + kx) * Cout + n) * Cin + c];
      } else {
        v = w[((size_t)(ky * KW + kx) * Cin + c) * Cout + n];
      }
    }
    const unsigned short h = f2bf(v);
    hi[i] = h;
    if (lo != nullptr) lo[i] = f2bf(v - bf2f(h));
  }
}

int conv_ksteps(int KH, int KW, int Cin) {
  if (Cin <= 8) return cdiv(KH * KW, 4);
  return KH * KW * (Cin / 32);
}

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

struct TileCfg { int wm, wn, mi, ni, tw; };

template <int WM, int WN, int MI, int NI, int TW, bool NARROW, bool PRECISE>
int launch_conv(ConvKArgs& a, hipStream_t stream) {
  constexpr int BM = WM * MI * 16, BN = WN * NI * 16, TH = BM / TW;
  a.tiles_x = cdiv(a.Wo, TW);
  a.tiles_y = cdiv(a.Ho, TH);
  a.nblocks = cdiv(a.Cout, BN);
  a.HT = (TH - 1) * a.stride + a.KH;
  a.WT = (TW - 1) * a.stride + a.KW;
  a.NPIX = a.HT * a.WT;
  a.NPIXP = roundup(a.NPIX, 16) + (a.stride == 2 ? 1 : 0);
  a.wt_magic = ((1 << 24) + a.WT - 1) / a.WT;
  a.ntaps = a.KH * a.KW;
  const int bplanes = PRECISE ? 2 : 1;
  const int b_bytes = 2 * bplanes * KC * 4 * BN * 16;
  const int misc = 2 * a.Cin * 4 + roundup(a.ntaps, 4) * 4 + WM * BN * 2 * 4;
  const int budget = 160 * 1024 - b_bytes - roundup(misc, 16) - 64;
  if (NARROW) {
    a.cgs = 8; a.ngroups = 1; a.log2nq = 0; a.log2cbg = 0;
    a.ksg = cdiv(a.ntaps, 4);
  } else {
    int cgs = a.Cin;
    while (cgs > 32 && (cgs / 8) * a.NPIXP * 16 * bplanes > budget) cgs >>= 1;
    if ((cgs / 8) * a.NPIXP * 16 * bplanes > budget || (a.Cin % cgs) != 0) return HDRSKY_EUNSUPPORTED;
    a.cgs = cgs; a.ngroups = a.Cin / cgs; a.log2nq = ilog2(cgs / 8); a.log2cbg = ilog2(cgs / 32);
    if ((1 << a.log2nq) != cgs / 8) return HDRSKY_EUNSUPPORTED;
    a.ksg = a.ntaps * (cgs / 32);
  }
  const int a_plane = (NARROW ? 1 : a.cgs / 8) * a.NPIXP * 16;
  a.off_alo = a_plane;
  a.off_b = a_plane * bplanes;
  a.off_ss = a.off_b + b_bytes;
  a.off_tap = a.off_ss + 2 * a.Cin * 4;
  a.off_stat = roundup(a.off_tap + a.ntaps * 4, 16);
  const int lds = a.off_stat + WM * BN * 2 * 4;
  if (lds > 160 * 1024) return HDRSKY_EUNSUPPORTED;
  auto kern = conv_igemm_kernel<WM, WN, MI, NI, TW, NARROW, PRECISE>;
  static int max_lds_set = 0;
  if (lds > max_lds_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return HDRSKY_ELAUNCH;
    max_lds_set = 160 * 1024;
  }
  const int grid = a.B * a.tiles_y * a.tiles_x * a.nblocks;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

template <bool NARROW, bool PRECISE>
int dispatch_tile(ConvKArgs& a, const TileCfg& t, hipStream_t s) {
#define HDRSKY_CASE(WM_, WN_, MI_, NI_, TW_)                                              \
  if (t.wm == WM_ && t.wn == WN_ && t.mi == MI_ && t.ni == NI_ && t.tw == TW_)           \
    return launch_conv<WM_, WN_, MI_, NI_, TW_, NARROW, PRECISE>(a, s);
  // BN = 64
  HDRSKY_CASE(2, 2, 4, 2, 32) HDRSKY_CASE(2, 2, 2, 2, 32) HDRSKY_CASE(2, 2, 2, 2, 16)
  // BN = 32
  HDRSKY_CASE(4, 1, 4, 2, 32) HDRSKY_CASE(4, 1, 2, 2, 32) HDRSKY_CASE(2, 2, 2, 1, 32) HDRSKY_CASE(2, 2, 2, 1, 16)
  // BN = 16
  HDRSKY_CASE(4, 1, 4, 1, 32) HDRSKY_CASE(4, 1, 2, 1, 32) HDRSKY_CASE(4, 1, 1, 1, 16)
#undef HDRSKY_CASE
  return HDRSKY_EUNSUPPORTED;
}

// Tile heuristic: widest N block the layer fills, then the largest pixel tile that still
// yields >= ~1 workgroup per CU (256 CUs), preferring more workgroups for small problems.
TileCfg choose_tile(const hdrsky_conv_desc* d) {
  const int bn = d->Cout >= 64 ? 64 : (d->Cout >= 32 ? 32 : 16);
  const int tw = (d->Wo >= 32) ? 32 : 16;
  const int nblk = cdiv(d->Cout, bn);
  auto wgs = [&](int bm) { return d->B * cdiv(d->Ho, bm / tw) * cdiv(d->Wo, tw) * nblk; };
  TileCfg t;
  t.tw = tw;
  if (bn == 64) {
    if (tw == 32 && wgs(128) >= 256) t = TileCfg{2, 2, 4, 2, 32};
    else t = TileCfg{2, 2, 2, 2, tw};
  } else if (bn == 32) {
    if (tw == 32 && wgs(256) >= 256) t = TileCfg{4, 1, 4, 2, 32};
    else if (tw == 32 && wgs(128) >= 256) t = TileCfg{4, 1, 2, 2, 32};
    else t = TileCfg{2, 2, 2, 1, tw};
  } else {
    if (tw == 32 && wgs(256) >= 256) t = TileCfg{4, 1, 4, 1, 32};
    else if (tw == 32) t = TileCfg{4, 1, 2, 1, 32};
    else t = TileCfg{4, 1, 1, 1, 16};
  }
  return t;
}

}  // namespace

extern "C" {

const char* hdrsky_version(void) { return "hdrsky 0.1 (gfx950)"; }

int hdrsky_conv_desc_init(hdrsky_conv_desc* d, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                          int same, int upsample) {
  if (!d || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return HDRSKY_EINVAL;
  if (stride != 1 && stride != 2) return HDRSKY_EINVAL;
  if (upsample != 1 && upsample != 2) return HDRSKY_EINVAL;
  *d = hdrsky_conv_desc{};
  d->B = B; d->H = H; d->W = W; d->Cin = Cin; d->Cout = Cout;
  d->KH = KH; d->KW = KW; d->stride = stride; d->upsample = upsample; d->dilate = 1;
  d->Hc = H * upsample; d->Wc = W * upsample;
  if (same) {
    d->Ho = cdiv(d->Hc, stride); d->Wo = cdiv(d->Wc, stride);
    const int th = (d->Ho - 1) * stride + KH - d->Hc, tw = (d->Wo - 1) * stride + KW - d->Wc;
    d->pad_t = (th > 0 ? th : 0) / 2; d->pad_l = (tw > 0 ? tw : 0) / 2;
  } else {
    if (d->Hc < KH || d->Wc < KW) return HDRSKY_EINVAL;
    d->Ho = (d->Hc - KH) / stride + 1; d->Wo = (d->Wc - KW) / stride + 1;
    d->pad_t = d->pad_l = 0;
  }
  d->compute = HDRSKY_BF16;
  d->in_mode = HDRSKY_IN_NONE; d->in_slope = 1.f; d->out_slope = 1.f; d->in_eps = 1e-3f;
  return HDRSKY_OK;
}

size_t hdrsky_conv_packed_elems(int KH, int KW, int Cin, int Cout) {
  return (size_t)conv_ksteps(KH, KW, Cin) * 4 * roundup(Cout, 64) * 8;
}

int hdrsky_conv_pack_weights(const float* w, int KH, int KW, int Cin, int Cout, int transpose_flip, void* packed_hi,
                             void* packed_lo, void* stream) {
  if (!w || !packed_hi) return HDRSKY_EINVAL;
  if (Cin > 8 && (Cin % 32) != 0) return HDRSKY_EUNSUPPORTED;
  const int ks = conv_ksteps(KH, KW, Cin);
  const int Npad = roundup(Cout, 64);
  const size_t total = (size_t)ks * 4 * Npad * 8;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, KH, KW, Cin, Cout, Npad,
                     Cin <= 8 ? 1 : 0, transpose_flip, ks, (unsigned short*)packed_hi, (unsigned short*)packed_lo);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_conv_stats_nparts(const hdrsky_conv_desc* d) {
  if (!d) return HDRSKY_EINVAL;
  const TileCfg t = choose_tile(d);
  const int bm = t.wm * t.mi * 16;
  return cdiv(d->Ho, bm / t.tw) * cdiv(d->Wo, t.tw);
}

int hdrsky_conv2d_fwd(const hdrsky_conv_desc* d, const float* x, const void* w_hi, const void* w_lo,
                      const float* bias, const float* in_scale, const float* in_shift, const float* in_part,
                      const float* in_gamma, const float* in_beta, const float* residual, float* y,
                      float* stats_part, void* stream) {
  if (!d || !x || !w_hi || !y) return HDRSKY_EINVAL;
  const bool narrow = d->Cin <= 8;
  if (!narrow && (d->Cin % 32) != 0) return HDRSKY_EUNSUPPORTED;
  if (narrow && (d->upsample != 1 || d->dilate != 1)) return HDRSKY_EUNSUPPORTED;
  if (d->upsample == 2 && d->dilate == 2) return HDRSKY_EINVAL;
  const bool precise = d->compute == HDRSKY_BF16X3;
  if (precise && !w_lo) return HDRSKY_EINVAL;
  if (d->in_mode == HDRSKY_IN_AFFINE && (!in_scale || !in_shift)) return HDRSKY_EINVAL;
  if (d->in_mode == HDRSKY_IN_PARTIALS && (!in_part || !in_gamma || !in_beta || d->in_nparts <= 0)) return HDRSKY_EINVAL;
  if (d->want_stats && !stats_part) return HDRSKY_EINVAL;
  ConvKArgs a{};
  a.x = x; a.whi = (const uint4*)w_hi; a.wlo = (const uint4*)w_lo; a.bias = bias;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_part = in_part; a.in_gamma = in_gamma; a.in_beta = in_beta;
  a.residual = residual; a.y = y; a.stats = stats_part;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.Npad = roundup(d->Cout, 64);
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.upsample = d->upsample; a.dilate = d->dilate; a.Hc = d->Hc; a.Wc = d->Wc;
  a.in_mode = d->in_mode; a.ss_bstride = d->ss_bstride; a.in_nparts = d->in_nparts;
  a.in_eps = d->in_eps; a.in_inv_count = 1.f / (float)(d->H * d->W); a.in_slope = d->in_slope;
  a.out_slope = d->out_slope; a.final_relu = d->final_relu; a.want_stats = d->want_stats;
  const TileCfg t = choose_tile(d);
  hipStream_t s = (hipStream_t)stream;
  if (narrow) return precise ? dispatch_tile<true, true>(a, t, s) : dispatch_tile<true, false>(a, t, s);
  return precise ? dispatch_tile<false, true>(a, t, s) : dispatch_tile<false, false>(a, t, s);
}

}  // extern "C"
