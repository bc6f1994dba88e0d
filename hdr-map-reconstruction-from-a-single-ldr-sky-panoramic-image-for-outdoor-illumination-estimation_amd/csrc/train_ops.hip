// Training-step kernels that are not convolutions (train.py:301-415): BatchNorm train-mode statistics and
// backward, activation / pooling backward, bilinear-resize adjoint, Dense weight gradient, soft-max backward,
// the losses with their gradients (KL, L1 means, LSGAN, DoG), blend / decoder-tail / sun-radiance backward and
// the fused multi-tensor RMSprop.  fp32 throughout; scalar loss values are accumulated with atomics (they are
// logging only - no gradient depends on them); every gradient is deterministic.
#include <atomic>

#include "common.h"
#include "hooks.h"

namespace {

constexpr float LN11 = 2.3978953f;

// block-wide sum (256 threads) -> one atomicAdd per block
__device__ __forceinline__ void block_atomic_add(float v, float scale, float* dst) {
  __shared__ float sblk[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sblk[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0 && dst) atomicAdd(dst, ((sblk[0] + sblk[1]) + (sblk[2] + sblk[3])) * scale);
}


// ------------------------------------------------------------------------------------------------------------
// Keras BatchNormalization, training mode (discriminator.py:25, sunrad_net.py:26): batch mean / biased variance
// from the producing conv's per-tile partials; moving stats <- 0.99*moving + 0.01*batch (variance Bessel-corrected).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) bn_train_finalize_kernel(const float* __restrict__ part, int nparts_total, int C,
                                                               float count, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float momentum,
                                                               float* moving_mean, float* moving_var, float* mean,
                                                               float* rstd, float* scale, float* shift, int rows) {
  const int c = blockIdx.x;  // one wave per channel; lanes stride over the partials (fixed shuffle tree: deterministic)
  float s = 0.f, ss = 0.f;
  for (int p = threadIdx.x; p < nparts_total; p += 64) { s += part[(size_t)(2 * p) * C + c]; ss += part[(size_t)(2 * p + 1) * C + c]; }
  s = wave_sum(s); ss = wave_sum(ss);
  const float m = s / count;
  const float var = fmaxf(ss / count - m * m, 0.f);
  const float r = 1.f / sqrtf(var + eps);
  const float inv = gamma[c] * r;
  // scale / shift as `rows` identical rows of a [rows][C] table (per-sample affine tables of a batch that holds several
  // BatchNorm groups: the paired discriminator passes)
  for (int q = threadIdx.x; q < rows; q += 64) { scale[(size_t)q * C + c] = inv; shift[(size_t)q * C + c] = beta[c] - m * inv; }
  if (threadIdx.x != 0) return;
  mean[c] = m; rstd[c] = r;
  if (moving_mean) {
    moving_mean[c] = moving_mean[c] * momentum + m * (1.f - momentum);
    moving_var[c] = moving_var[c] * momentum + var * (count / fmaxf(count - 1.f, 1.f)) * (1.f - momentum);
  }
}

// four consecutive elements (element offset e, a multiple of 4) of an incoming gradient stored as fp32 or as bf16 (a
// data-gradient conv's y_bf16 output whose only reader is the kernel at hand)
__device__ __forceinline__ float4 ld4grad(const float* dy, int dy16, size_t e) { return ld4any(dy, dy16, e); }

// g = dy * act'(pre), pre = xhat*gamma+beta, xhat = (x-mean[c])*rstd[c]; per-block partial (sum g, sum g*xhat)
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float slope, size_t npix, int C, float* __restrict__ part,
                                                            int flags) {
  const int dy16 = flags & 2, x16 = flags & 4;     // storage of dy / of x: bf16
  extern __shared__ float sm[];  // [256][8] partials
  const int c4 = C >> 2;
  const int lanes_c = c4 < 256 ? c4 : 256;            // threads along channels (C/4 <= 256)
  const int rows = 256 / lanes_c;
  const int tc = threadIdx.x % lanes_c, tr = threadIdx.x / lanes_c;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (tr < rows) {
    const int c = tc * 4;
    float mu[4], rs[4], gm[4], bt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; rs[j] = rstd[c + j]; gm[j] = gamma[c + j]; bt[j] = beta[c + j]; }
    for (size_t p = (size_t)blockIdx.x * rows + tr; p < npix; p += (size_t)gridDim.x * rows) {
      const float4 xv = ld4any(x, x16, p * C + c);
      const float4 dv = ld4grad(dy, dy16, p * C + c);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (xs[j] - mu[j]) * rs[j];
        const float g = ds[j] * ((xh * gm[j] + bt[j]) > 0.f ? 1.f : slope);
        s1[j] += g; s2[j] += g * xh;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { sm[threadIdx.x * 8 + j] = s1[j]; sm[threadIdx.x * 8 + 4 + j] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < lanes_c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += sm[(r * lanes_c + threadIdx.x) * 8 + j];
      part[((size_t)blockIdx.x * 2 + (j >> 2)) * C + threadIdx.x * 4 + (j & 3)] = t;
    }
  }
}

// sums over blocks -> dbeta, dgamma (accumulated into the gradient buffers) and the two means for the apply pass
__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, int C,
                                                             float count, float* m1m2, float* dgamma, float* dbeta) {
  const int c = blockIdx.x;  // one wave per channel
  float s1 = 0.f, s2 = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 64) { s1 += part[((size_t)b * 2) * C + c]; s2 += part[((size_t)b * 2 + 1) * C + c]; }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (threadIdx.x != 0) return;
  m1m2[c] = s1 / count; m1m2[C + c] = s2 / count;
  if (dbeta) dbeta[c] += s1;
  if (dgamma) dgamma[c] += s2;
}

// The same for a batch that is spread over several replicas (SyncBN, parallel.py sync_batch_stats): the two means of the
// apply pass run over the partial blocks of EVERY replica (part_all, gathered; count_all = the global pixel count), while
// the parameter gradients receive this replica's blocks only (the gradient exchange sums them over the replicas)
__global__ void __launch_bounds__(64) bn_bwd_finalize2_kernel(const float* __restrict__ part_all, int nblocks_all, float count_all,
                                                              const float* __restrict__ part_local, int nblocks_local, int C,
                                                              float* m1m2, float* dgamma, float* dbeta) {
  const int c = blockIdx.x;
  float s1 = 0.f, s2 = 0.f, l1 = 0.f, l2 = 0.f;
  for (int b = threadIdx.x; b < nblocks_all; b += 64) { s1 += part_all[((size_t)b * 2) * C + c]; s2 += part_all[((size_t)b * 2 + 1) * C + c]; }
  for (int b = threadIdx.x; b < nblocks_local; b += 64) { l1 += part_local[((size_t)b * 2) * C + c]; l2 += part_local[((size_t)b * 2 + 1) * C + c]; }
  s1 = wave_sum(s1); s2 = wave_sum(s2); l1 = wave_sum(l1); l2 = wave_sum(l2);
  if (threadIdx.x != 0) return;
  m1m2[c] = s1 / count_all; m1m2[C + c] = s2 / count_all;
  if (dbeta) dbeta[c] += l1;
  if (dgamma) dgamma[c] += l2;
}

// four consecutive elements of a gradient tensor, fp32 or bf16 (element group i4)
__device__ __forceinline__ void st4(void* p, int as_bf16, size_t i4, const float (&o)[4]) {
  if (as_bf16)
    reinterpret_cast<uint2*>(p)[i4] = uint2{(unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16), (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16)};
  else
    reinterpret_cast<float4*>(p)[i4] = make_float4(o[0], o[1], o[2], o[3]);
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                    const float* __restrict__ m1m2, size_t n4, int C, void* __restrict__ dx, int dx_bf16) {
  const int c4 = C >> 2;
  const int dy16 = dx_bf16 & 2, x16 = dx_bf16 & 4;      // bit 1: dy given as bf16; bit 2: x stored as bf16
  dx_bf16 &= 1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    const float4 xv = ld4any(x, x16, i * 4);
    const float4 dv = ld4grad(dy, dy16, i * 4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xs[j] - mean[c + j]) * rstd[c + j];
      const float g = ds[j] * ((xh * gamma[c + j] + beta[c + j]) > 0.f ? 1.f : slope);
      o[j] = gamma[c + j] * rstd[c + j] * (g - m1m2[c + j] - xh * m1m2[C + c + j]);
    }
    st4(dx, dx_bf16, i, o);
  }
}

// dx = dy * act'(x*scale[c]+shift[c]) * scale[c]   (BatchNorm in inference mode = constant per-channel affine;
// scale == nullptr: plain activation backward on the ACTIVATED tensor y passed as x: dx = dy * (y > 0 ? 1 : slope))
template <int V>   // V = 4: float4 per thread (C % 4 == 0, 16-byte aligned tensors), else 1
__global__ void affine_act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                      const float* __restrict__ scale, const float* __restrict__ shift, float slope,
                                      size_t n, int C, void* __restrict__ dxv, int dx_bf16) {
  float* dx = reinterpret_cast<float*>(dxv);
  const int dy16 = dx_bf16 & 2, x16 = dx_bf16 & 4;      // bit 1: dy given as bf16, bit 2: x stored as bf16 (V == 4 only)
  dx_bf16 &= 1;
  for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * blockDim.x * V) {
    const int c = (int)(i % C);
    float v[V], g[V], o[V];
    if (V == 4) {
      const float4 a = ld4any(x, x16, i), b = ld4grad(dy, dy16, i);
      v[0] = a.x; v[1 % V] = a.y; v[2 % V] = a.z; v[3 % V] = a.w;
      g[0] = b.x; g[1 % V] = b.y; g[2 % V] = b.z; g[3 % V] = b.w;
    } else {
      v[0] = x[i]; g[0] = dy[i];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      if (scale) o[e] = g[e] * ((v[e] * scale[c + e] + shift[c + e]) > 0.f ? 1.f : slope) * scale[c + e];
      else o[e] = g[e] * (v[e] > 0.f ? 1.f : slope);
    }
    if (V == 4) { const float o4[4] = {o[0], o[1 % V], o[2 % V], o[3 % V]}; st4(dxv, dx_bf16, i >> 2, o4); }
    else dx[i] = o[0];
  }
}

// ------------------------------------------------------------------------------------------------------------
// 2x2/2 max-pool (vgg16.py:85-86) and its backward fused with the ReLU in front of it: y is the post-ReLU conv
// output; the gradient goes to the first arg-max of each window and only where y > 0.
// ------------------------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ y, int B, int H, int W, int C, float* __restrict__ p) {
  const int c4 = C >> 2, Hp = H >> 1, Wp = W >> 1;
  const size_t total = (size_t)B * Hp * Wp * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4);
    const size_t pix = i / c4;
    const int pw = (int)(pix % Wp), ph = (int)((pix / Wp) % Hp), b = (int)(pix / ((size_t)Wp * Hp));
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = reinterpret_cast<const float4*>(y)[((size_t)(b * H + 2 * ph + (k >> 1)) * W + 2 * pw + (k & 1)) * c4 + cq];
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    reinterpret_cast<float4*>(p)[i] = m;
  }
}

__global__ void maxpool_relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dp, int B, int H, int W,
                                        int C, float* __restrict__ dy) {
  const int c4 = C >> 2, Hp = H >> 1, Wp = W >> 1;
  const size_t total = (size_t)B * Hp * Wp * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4);
    const size_t pix = i / c4;
    const int pw = (int)(pix % Wp), ph = (int)((pix / Wp) % Hp), b = (int)(pix / ((size_t)Wp * Hp));
    const float4 up = reinterpret_cast<const float4*>(dp)[i];
    const float us[4] = {up.x, up.y, up.z, up.w};
    float v[4][4];
    size_t idx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      idx[k] = ((size_t)(b * H + 2 * ph + (k >> 1)) * W + 2 * pw + (k & 1)) * c4 + cq;
      const float4 t = reinterpret_cast<const float4*>(y)[idx[k]];
      v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
    }
    float o[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int am = 0;
      float best = v[0][j];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][j] > best) { best = v[k][j]; am = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][j] = (k == am && v[k][j] > 0.f) ? us[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<float4*>(dy)[idx[k]] = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
  }
}

// ---- the same three on bf16 activations (the VGG16 chain in HDRSKY_BF16 mode) --------------------------------------
__device__ __forceinline__ void ld4bf(const unsigned short* p, size_t i4, float (&v)[4]) {   // element group i4 (4 bf16)
  const uint2 u = reinterpret_cast<const uint2*>(p)[i4];
  v[0] = __builtin_bit_cast(float, u.x << 16); v[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
  v[2] = __builtin_bit_cast(float, u.y << 16); v[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
}

__global__ void maxpool_fwd_bf16_kernel(const unsigned short* __restrict__ y, int B, int H, int W, int C,
                                        float* __restrict__ p32, unsigned short* __restrict__ p16) {
  const int c4 = C >> 2, Hp = H >> 1, Wp = W >> 1;
  const size_t total = (size_t)B * Hp * Wp * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4);
    const size_t pix = i / c4;
    const int pw = (int)(pix % Wp), ph = (int)((pix / Wp) % Hp), b = (int)(pix / ((size_t)Wp * Hp));
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v[4];
      ld4bf(y, ((size_t)(b * H + 2 * ph + (k >> 1)) * W + 2 * pw + (k & 1)) * c4 + cq, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
    }
    if (p32) reinterpret_cast<float4*>(p32)[i] = make_float4(m[0], m[1], m[2], m[3]);
    if (p16)      // the maximum of bf16 values is one of them: exact
      reinterpret_cast<uint2*>(p16)[i] = uint2{(unsigned)f2bf(m[0]) | ((unsigned)f2bf(m[1]) << 16),
                                               (unsigned)f2bf(m[2]) | ((unsigned)f2bf(m[3]) << 16)};
  }
}

// L1: dp is not read as given - it is (dp or 0) + the gradient of the L1 term wl * mean|pool - target| of this block's pooled
// features, whose value is accumulated into *loss (the perceptual term of train.py:308-313: three hdrsky_l1 launches per VGG16
// backward pass folded into the pool backward that consumed their output; same values: (+-wg/n) + dp)
template <bool OB, bool L1>
__global__ void __launch_bounds__(256) maxpool_relu_bwd_bf16_kernel(const unsigned short* __restrict__ y, const float* __restrict__ dp, int B,
                                                                    int H, int W, int C, void* __restrict__ dyv,
                                                                    const float* __restrict__ pool, const float* __restrict__ target,
                                                                    float gp, float lscale, float* loss) {
  const int c4 = C >> 2, Hp = H >> 1, Wp = W >> 1;
  const size_t total = (size_t)B * Hp * Wp * c4;
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4);
    const size_t pix = i / c4;
    const int pw = (int)(pix % Wp), ph = (int)((pix / Wp) % Hp), b = (int)(pix / ((size_t)Wp * Hp));
    float us[4];
    if (L1) {
      const float4 pa = reinterpret_cast<const float4*>(pool)[i], pb = reinterpret_cast<const float4*>(target)[i];
      const float d[4] = {pa.x - pb.x, pa.y - pb.y, pa.z - pb.z, pa.w - pb.w};
      acc += (fabsf(d[0]) + fabsf(d[1])) + (fabsf(d[2]) + fabsf(d[3]));
#pragma unroll
      for (int j = 0; j < 4; ++j) us[j] = d[j] > 0.f ? gp : (d[j] < 0.f ? -gp : 0.f);
      if (dp != nullptr) {
        const float4 q = reinterpret_cast<const float4*>(dp)[i];
        us[0] += q.x; us[1] += q.y; us[2] += q.z; us[3] += q.w;
      }
    } else {
      const float4 up = reinterpret_cast<const float4*>(dp)[i];
      us[0] = up.x; us[1] = up.y; us[2] = up.z; us[3] = up.w;
    }
    float v[4][4];
    size_t idx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      idx[k] = ((size_t)(b * H + 2 * ph + (k >> 1)) * W + 2 * pw + (k & 1)) * c4 + cq;
      ld4bf(y, idx[k], v[k]);
    }
    float o[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int am = 0;
      float best = v[0][j];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][j] > best) { best = v[k][j]; am = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][j] = (k == am && v[k][j] > 0.f) ? us[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (OB)
        reinterpret_cast<uint2*>(dyv)[idx[k]] = uint2{(unsigned)f2bf(o[k][0]) | ((unsigned)f2bf(o[k][1]) << 16),
                                                      (unsigned)f2bf(o[k][2]) | ((unsigned)f2bf(o[k][3]) << 16)};
      else
        reinterpret_cast<float4*>(dyv)[idx[k]] = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
    }
  }
  if (L1) block_atomic_add(acc, lscale, loss);
}

__global__ void act_bwd_bf16_kernel(const unsigned short* __restrict__ y, const float* __restrict__ dy, float slope, size_t n4,
                                    void* __restrict__ dx, int dx_bf16) {
  const int dy16 = dx_bf16 & 2;      // bit 1: dy given as bf16
  dx_bf16 &= 1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float v[4];
    ld4bf(y, i, v);
    const float4 g = ld4grad(dy, dy16, i * 4);
    const float o[4] = {g.x * (v[0] > 0.f ? 1.f : slope), g.y * (v[1] > 0.f ? 1.f : slope),
                        g.z * (v[2] > 0.f ? 1.f : slope), g.w * (v[3] > 0.f ? 1.f : slope)};
    st4(dx, dx_bf16, i, o);
  }
}

// ------------------------------------------------------------------------------------------------------------
// bilinear 2x resize (half-pixel centres) forward on an optional difference a-b, and its adjoint
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void axis2x(int o, int n, int& lo, int& hi, float& t) {
  const float s = (o + 0.5f) * 0.5f - 0.5f;
  const float f = floorf(s);
  lo = max((int)f, 0); hi = min((int)ceilf(s), n - 1); t = s - f;
}

__global__ void up2x_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int H, int W, int C,
                                float* __restrict__ y) {
  const size_t total = (size_t)B * 4 * H * W * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t pix = i / C;
    const int ox = (int)(pix % (2 * W)), oy = (int)((pix / (2 * W)) % (2 * H)), bb = (int)(pix / ((size_t)4 * W * H));
    int ylo, yhi, xlo, xhi; float ty, tx;
    axis2x(oy, H, ylo, yhi, ty); axis2x(ox, W, xlo, xhi, tx);
    auto at = [&](int yy, int xx) {
      const size_t k = ((size_t)(bb * H + yy) * W + xx) * C + c;
      return b ? a[k] - b[k] : a[k];
    };
    const float tl = at(ylo, xlo), tr = at(ylo, xhi), bl = at(yhi, xlo), br = at(yhi, xhi);
    const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
    y[i] = top + (bot - top) * ty;
  }
}

// The operand of a resize-deconvolution (ops.py:44-126: tf.image.resize 2x, then the conv), materialised once as bf16:
// y = bf16( resize2x( leaky( IN(x) ) ) ) with the InstanceNorm affine built from the producing conv's statistics partials
// (part == nullptr: x is already an activation).  Exactly the arithmetic - and the operation order - of the conv kernel's
// fused-upsample staging (conv_igemm.hip), so a plain conv on y returns what the fused one returns; the plain conv and
// the plain weight gradient are the faster kernels (no four-source blend per staged element), and the two decoders
// share the upsampled encoder output.  Block = (sample, run of output pixels); thread = (pixel, 8 channels).
__global__ void __launch_bounds__(256) up2x_xf_bf16_kernel(const float* __restrict__ x, int B, int H, int W, int C,
                                                           const float* __restrict__ part, int nparts,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float slope, uint4* __restrict__ y, int blocks_per_sample,
                                                           int x16, const float* __restrict__ gamma2 = nullptr,
                                                           const float* __restrict__ beta2 = nullptr, int gsplit = 0) {
  __shared__ float sSc[512], sSh[512];
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  const bool xf = part != nullptr;
  if (gsplit > 0 && b >= gsplit) { gamma = gamma2; beta = beta2; }      // paired tensors: the second half is another layer's output
  if (xf) {
    const float inv_count = 1.f / (float)(H * W);
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f, ss = 0.f;
      const float* pp = part + (size_t)b * nparts * 2 * C + c;
      for (int p = 0; p < nparts; ++p) { s += pp[(2 * p) * C]; ss += pp[(2 * p + 1) * C]; }
      const float mean = s * inv_count;
      const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
      const float inv = gamma[c] / sqrtf(var + eps);
      sSc[c] = inv; sSh[c] = beta[c] - mean * inv;
    }
    __syncthreads();
  }
  const int nq = C >> 3, OW = 2 * W, OH = 2 * H;
  const int nitems = OH * OW * nq;
  const int per = (nitems + blocks_per_sample - 1) / blocks_per_sample;
  const int i1 = min(nitems, (blk + 1) * per);
  for (int i = blk * per + threadIdx.x; i < i1; i += 256) {
    const int q = i % nq, pix = i / nq;
    const int cx = pix % OW, cy = pix / OW;
    const float sy = (cy + 0.5f) * 0.5f - 0.5f, sx = (cx + 0.5f) * 0.5f - 0.5f;
    const float fy = floorf(sy), fx = floorf(sx);
    const int ylo = min(max((int)fy, 0), H - 1), yhi = max(min((int)ceilf(sy), H - 1), 0);
    const int xlo = min(max((int)fx, 0), W - 1), xhi = max(min((int)ceilf(sx), W - 1), 0);
    const float ly = sy - fy, lx = sx - fx;
    const size_t xo = (size_t)b * H * W * C + q * 8;
    float4 t[8];
    ld8any(x, x16, xo + ((size_t)ylo * W + xlo) * C, t[0], t[1]);
    ld8any(x, x16, xo + ((size_t)ylo * W + xhi) * C, t[2], t[3]);
    ld8any(x, x16, xo + ((size_t)yhi * W + xlo) * C, t[4], t[5]);
    ld8any(x, x16, xo + ((size_t)yhi * W + xhi) * C, t[6], t[7]);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sc = xf ? sSc[q * 8 + j] : 1.f, sh = xf ? sSh[q * 8 + j] : 0.f;
      const int w = j >> 2, e = j & 3;
      const float tl = leaky(((const float*)&t[0 + w])[e] * sc + sh, slope);
      const float tr = leaky(((const float*)&t[2 + w])[e] * sc + sh, slope);
      const float bl = leaky(((const float*)&t[4 + w])[e] * sc + sh, slope);
      const float br = leaky(((const float*)&t[6 + w])[e] * sc + sh, slope);
      const float top = tl + (tr - tl) * lx;
      const float bot = bl + (br - bl) * lx;
      v[j] = top + (bot - top) * ly;
    }
    uint4 hi, lo;
    pack8<false>(v, hi, lo);
    y[(size_t)b * nitems + i] = hi;
  }
}

// dx[b,iy,ix,c] = sum over the <= 4x4 outputs that sample (iy,ix) of their bilinear weight * dy   (exact adjoint)
// V = channels per thread (4: float4 loads when C % 4 == 0; 1: the 3-channel DoG images)
template <int V>
__global__ void __launch_bounds__(256) up2x_bwd_kernel(const float* __restrict__ dy, int B, int H, int W, int C, float scale,
                                                       int accumulate, float* __restrict__ dx) {
  // bit 1 of `accumulate`: dy is GIVEN as bf16 (the data-gradient conv of a resize-deconvolution writes the gradient at the
  // doubled resolution - 67 MB in fp32 for the 64-channel layer of a batch of 32 - for this launch alone to read)
  const bool dy16 = (accumulate & 2) != 0;
  // bit 2: PAIRED gradient - dy holds 2 B samples and dx[b] = adjoint(dy[b]) + adjoint(dy[b + B]): the two decoders' gradients
  // with respect to the encoder output they share, in the order of two accumulating calls (first half, then second half)
  const bool pair = (accumulate & 4) != 0;
  accumulate &= 1;
  const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy);
  const int CV = C / V;
  const size_t total = (size_t)B * H * W * CV;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V;
    const size_t pix = i / CV;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), bb = (int)(pix / ((size_t)W * H));
    float wy[4], wx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oy = 2 * iy - 1 + k, ox = 2 * ix - 1 + k;
      wy[k] = 0.f; wx[k] = 0.f;
      if (oy >= 0 && oy < 2 * H) {
        int lo, hi; float t; axis2x(oy, H, lo, hi, t);
        wy[k] = (lo == iy ? 1.f - t : 0.f) + (hi == iy ? t : 0.f);
      }
      if (ox >= 0 && ox < 2 * W) {
        int lo, hi; float t; axis2x(ox, W, lo, hi, t);
        wx[k] = (lo == ix ? 1.f - t : 0.f) + (hi == ix ? t : 0.f);
      }
    }
    float s[V], s2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { s[v] = 0.f; s2[v] = 0.f; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
    if (half == 1 && !pair) break;
    const int bs = bb + half * B;       // sample of dy
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float w = wy[ky] * wx[kx];
        if (w != 0.f) {   // zero weight <=> output pixel outside the image: never dereferenced
          const size_t off = ((size_t)(bs * 2 * H + 2 * iy - 1 + ky) * 2 * W + 2 * ix - 1 + kx) * C + c;
          const float* p = dy + off;
          if (V == 4) {
            float4 d;
            if (dy16) {
              const uint2 u = *reinterpret_cast<const uint2*>(dyh + off);
              d = make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                              __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
            } else {
              d = *reinterpret_cast<const float4*>(p);
            }
            if (half == 0) { s[0] += w * d.x; s[1 % V] += w * d.y; s[2 % V] += w * d.z; s[3 % V] += w * d.w; }
            else { s2[0] += w * d.x; s2[1 % V] += w * d.y; s2[2 % V] += w * d.z; s2[3 % V] += w * d.w; }
          } else {
            if (half == 0) s[0] += w * (dy16 ? bf2f(dyh[off]) : p[0]);
            else s2[0] += w * (dy16 ? bf2f(dyh[off]) : p[0]);
          }
        }
      }
    }
    float* o = dx + pix * C + c;
    if (V == 4) {
      float4 r = make_float4(s[0] * scale, s[1 % V] * scale, s[2 % V] * scale, s[3 % V] * scale);
      if (accumulate) { const float4 q = *reinterpret_cast<const float4*>(o); r.x = q.x + r.x; r.y = q.y + r.y; r.z = q.z + r.z; r.w = q.w + r.w; }
      if (pair) { r.x += s2[0] * scale; r.y += s2[1 % V] * scale; r.z += s2[2 % V] * scale; r.w += s2[3 % V] * scale; }
      *reinterpret_cast<float4*>(o) = r;
    } else {
      float r = accumulate ? o[0] + s[0] * scale : s[0] * scale;
      if (pair) r += s2[0] * scale;
      o[0] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// DoG loss (tf_utils.py:61-73, train.py:316-322).  All operators are linear, so DoG(y) - DoG(t) = DoG(y - t):
//   e_up = resize2x(y - t)                      (up2x_fwd_kernel)
//   base = G(1.2489996) e_up                     (blur3_kernel)
//   d_i  = G(s_{i+1}) base - G(s_i) base,  i = 0..3, s = {1.2262735, 1.5450078, 1.9465878, 2.452547, 3.0900156}
//   loss = sum_i mean|d_i|;  h_j = g_{j-1} - g_j with g_i = sign(d_i)/N     (dog_mid_kernel)
//   d base = sum_j G(s_j)^T h_j                  (dog_mid_bwd_kernel),  then G^T (blur3 transpose) and resize adjoint.
// 3x3 Gaussian, TFA kernel softmax(-x^2/(2 sigma^2)), REFLECT padding (no edge repeat).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gauss3(float sigma, float& a, float& c) {
  const float e = expf(-1.f / (2.f * sigma * sigma));
  c = 1.f / (1.f + 2.f * e);
  a = e * c;
}
__device__ __forceinline__ int refl(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }

// forward blur (transpose = 0) or its adjoint (transpose = 1) on [B,H,W,C]
__global__ void blur3_kernel(const float* __restrict__ x, int B, int H, int W, int C, float sigma, int transpose,
                             float* __restrict__ y) {
  float a, c0;
  gauss3(sigma, a, c0);
  const size_t total = (size_t)B * H * W * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t pix = i / C;
    const int px = (int)(pix % W), py = (int)((pix / W) % H), bb = (int)(pix / ((size_t)W * H));
    const float* xb = x + (size_t)bb * H * W * C + c;
    float s = 0.f;
    if (!transpose) {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
          s += (dy ? a : c0) * (dx ? a : c0) * xb[((size_t)refl(py + dy, H) * W + refl(px + dx, W)) * C];
    } else {
      // adjoint: contributions of every output p whose (reflected) tap lands on this input position
      float wy[4], wx[4]; int ys[4], xs[4];
      ys[0] = py - 1; wy[0] = py - 1 >= 0 ? a : 0.f;
      ys[1] = py; wy[1] = c0;
      ys[2] = py + 1; wy[2] = py + 1 < H ? a : 0.f;
      ys[3] = py == 1 ? 0 : H - 1; wy[3] = (py == 1 ? a : 0.f) + ((py == H - 2 && py != 1) ? a : 0.f);
      if (py == 1 && py == H - 2) { /* H == 3: both mirrors hit row 1 from rows 0 and 2 */ }
      xs[0] = px - 1; wx[0] = px - 1 >= 0 ? a : 0.f;
      xs[1] = px; wx[1] = c0;
      xs[2] = px + 1; wx[2] = px + 1 < W ? a : 0.f;
      xs[3] = px == 1 ? 0 : W - 1; wx[3] = (px == 1 ? a : 0.f) + ((px == W - 2 && px != 1) ? a : 0.f);
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const float w = wy[ky] * wx[kx];
          if (w != 0.f) s += w * xb[((size_t)ys[ky] * W + xs[kx]) * C];
        }
    }
    y[i] = s;
  }
}

__constant__ float DOG_S[5] = {1.2262735f, 1.5450078f, 1.9465878f, 2.452547f, 3.0900156f};

// loss += weight * sum_i mean|d_i|;  h[j] (5 planes, each [n]) = g_{j-1} - g_j with g_i = weight*sign(d_i)/n
__global__ void __launch_bounds__(256) dog_mid_kernel(const float* __restrict__ base, int B, int H, int W, int C,
                                                      float weight, float* __restrict__ h, float* loss) {
  float a[5], c0[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) gauss3(DOG_S[j], a[j], c0[j]);
  const size_t total = (size_t)B * H * W * C;
  const float inv_n = 1.f / (float)total;
  float lacc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t pix = i / C;
    const int px = (int)(pix % W), py = (int)((pix / W) % H), bb = (int)(pix / ((size_t)W * H));
    const float* xb = base + (size_t)bb * H * W * C + c;
    float centre = 0.f, edge = 0.f, corner = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const float v = xb[((size_t)refl(py + dy, H) * W + refl(px + dx, W)) * C];
        if (dy == 0 && dx == 0) centre = v; else if (dy == 0 || dx == 0) edge += v; else corner += v;
      }
    float bl[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) bl[j] = c0[j] * c0[j] * centre + a[j] * c0[j] * edge + a[j] * a[j] * corner;
    float g[6];
    g[0] = 0.f; g[5] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = bl[k + 1] - bl[k];
      lacc += fabsf(d);
      g[k + 1] = d > 0.f ? weight * inv_n : (d < 0.f ? -weight * inv_n : 0.f);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) h[(size_t)j * total + i] = g[j] - g[j + 1];
  }
  block_atomic_add(lacc, inv_n, loss);
}

// d base = sum_j G(s_j)^T h_j
__global__ void dog_mid_bwd_kernel(const float* __restrict__ h, int B, int H, int W, int C, float* __restrict__ dbase) {
  float a[5], c0[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) gauss3(DOG_S[j], a[j], c0[j]);
  const size_t total = (size_t)B * H * W * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t pix = i / C;
    const int px = (int)(pix % W), py = (int)((pix / W) % H), bb = (int)(pix / ((size_t)W * H));
    // adjoint tap multiplicities along each axis: m[k] = how many times neighbour k contributes with weight `a`
    int ys[4], xs[4]; float my[4], mx[4];
    ys[0] = py - 1; my[0] = py - 1 >= 0 ? 1.f : 0.f;
    ys[1] = py; my[1] = -1.f;  // marker: centre weight
    ys[2] = py + 1; my[2] = py + 1 < H ? 1.f : 0.f;
    ys[3] = py == 1 ? 0 : H - 1; my[3] = (py == 1 ? 1.f : 0.f) + ((py == H - 2 && py != 1) ? 1.f : 0.f);
    xs[0] = px - 1; mx[0] = px - 1 >= 0 ? 1.f : 0.f;
    xs[1] = px; mx[1] = -1.f;
    xs[2] = px + 1; mx[2] = px + 1 < W ? 1.f : 0.f;
    xs[3] = px == 1 ? 0 : W - 1; mx[3] = (px == 1 ? 1.f : 0.f) + ((px == W - 2 && px != 1) ? 1.f : 0.f);
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        if (my[ky] == 0.f || mx[kx] == 0.f) continue;
        const size_t src = ((size_t)(bb * H + ys[ky]) * W + xs[kx]) * C + c;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const float wy = my[ky] < 0.f ? c0[j] : a[j] * my[ky];
          const float wx = mx[kx] < 0.f ? c0[j] : a[j] * mx[kx];
          s += wy * wx * h[(size_t)j * total + src];
        }
      }
    dbase[i] = s;
  }
}


// The whole DoG term - resize, base blur, five-level pyramid, L1, and the adjoint chain back to d y - in ONE launch: a workgroup
// owns RB low-resolution rows of a sample (full width) and carries the chain through LDS, recomputing the few halo rows every
// stage needs (up: +-5 high-resolution rows, base +-4, signs +-3, d base +-2, d up +-1).  The staged path above is seven
// launches of 12-28 us each on the training step's critical chain - per-element passes that spend their time on index
// arithmetic (64-bit divisions) and tap weights, not on their 6 MB of data.  Here a thread owns ONE COLUMN (pixel column x
// channel) of the band: everything that depends on the column - neighbour offsets under REFLECT padding, the multiplicities
// of the adjoint taps, the bilinear weights - is computed once, the row loop slides a three-row window down the column
// (three LDS reads per element and stage instead of nine), and what depends on the row is wave-uniform.
//   adjoint of a REFLECT-padded 3-tap filter along an axis of n points: point p collects from p-1, p, p+1 with the edge
//   weight times m(p-1 -> p), where the neighbour at index 0 / n-1 counts twice for p = 1 / n-2 and the neighbours -1 / n
//   do not exist (blur3_kernel's transpose branch, written as multiplicities).
// Between the pyramid and its adjoint only the SIGNS of the four differences survive: one byte per element (2 bits each),
// and sum_j G(s_j)^T h_j = sum_k sign(d_k) (G(s_k+1) - G(s_k))^T collapses, per tap type (centre / edge / corner), into a
// 256-entry table indexed by that byte.  Same operators as the staged path; the fp32 summation order differs.
struct DogArgs {
  const float* y; const float* t; float* loss; float* dy;
  int B, H, W, C, RB, nbands;
  int SW, nstrips;               // low-resolution columns per workgroup, strips per row band (1: full width)
  float weight;
  int offB, offE, offS, offL;    // LDS byte offsets: buffer B, the low-resolution difference rows, the sign bytes, the tables
};
__global__ void __launch_bounds__(1024) dog_fused_kernel(const DogArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  float* bufA = reinterpret_cast<float*>(dsm);            // up, later d base
  float* bufB = reinterpret_cast<float*>(dsm + g.offB);   // base, later d up
  float* sE = reinterpret_cast<float*>(dsm + g.offE);
  unsigned char* sS = dsm + g.offS;
  float* sLut = reinterpret_cast<float*>(dsm + g.offL);   // [3 tap types][256 sign bytes]
  const int tid = threadIdx.x, NT = 1024;
  const int strip = blockIdx.x % g.nstrips, bb_ = blockIdx.x / g.nstrips;
  const int b = bb_ / g.nbands, band = bb_ - b * g.nbands;
  const int H = g.H, W = g.W, C = g.C, H2 = 2 * H, W2 = 2 * W;
  const int r0 = band * g.RB, r1 = min(H, r0 + g.RB);
  // column strip: the workgroup's own low-resolution columns [c0, c1) and the high-resolution columns [xs0, xs1) it carries
  // through LDS (+-5: the same halo as along the rows - every stage needs one column more than the next); el0 / ew: the
  // low-resolution columns the resize of those reads.  One strip (the 32x128 maps): the whole row, no halo
  const int c0 = strip * g.SW, c1 = min(W, c0 + g.SW);
  const int xs0 = max(0, 2 * c0 - 5), xs1 = min(W2, 2 * c1 + 5), SWh = xs1 - xs0;
  int el0, el1;
  { int lo, hi; float tt; axis2x(xs0, W, lo, hi, tt); el0 = lo; axis2x(xs1 - 1, W, lo, hi, tt); el1 = hi + 1; }
  const int RW = SWh * C, rw = (el1 - el0) * C;
  const int Pd0 = max(0, 2 * r0 - 1), Pd1 = min(H2, 2 * r1 + 1);
  const int Pb0 = max(0, Pd0 - 1), Pb1 = min(H2, Pd1 + 1);
  const int Ps0 = max(0, Pb0 - 1), Ps1 = min(H2, Pb1 + 1);
  const int Pa0 = max(0, Ps0 - 1), Pa1 = min(H2, Ps1 + 1);
  const int Pu0 = max(0, Pa0 - 1), Pu1 = min(H2, Pa1 + 1);
  int e0, e1;
  { int lo, hi; float tt; axis2x(Pu0, H, lo, hi, tt); e0 = lo; axis2x(Pu1 - 1, H, lo, hi, tt); e1 = hi + 1; }
  float ga[5], gc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) gauss3(DOG_S[j], ga[j], gc[j]);
  float ba, bc;
  gauss3(1.2489996f, ba, bc);
  const float inv_n = 1.f / (float)((size_t)g.B * H2 * W2 * C);

  // ---- e = y - t on the low-resolution rows the band's resize reads; the sign-byte tables -----------------------------------
  for (int i = tid; i < (e1 - e0) * rw; i += NT) {
    const int er = i / rw, ej = i - er * rw;
    const size_t k = ((size_t)(b * H + e0 + er) * W + el0) * C + ej;
    sE[i] = g.y[k] - g.t[k];
  }
  if (tid < 256) {
    float u[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float sg = (float)((tid >> (2 * k)) & 3) - 1.f;        // sign(d_k): field value 0 / 1 / 2
      u[0] += sg * (gc[k + 1] * gc[k + 1] - gc[k] * gc[k]);         // centre tap
      u[1] += sg * (ga[k + 1] * gc[k + 1] - ga[k] * gc[k]);         // edge
      u[2] += sg * (ga[k + 1] * ga[k + 1] - ga[k] * ga[k]);         // corner
    }
    sLut[tid] = u[0]; sLut[256 + tid] = u[1]; sLut[512 + tid] = u[2];
  }
  // ---- the thread's column -------------------------------------------------------------------------------------------------
  // (X: the column in the IMAGE - reflections, multiplicities and bilinear weights follow the image's borders; the offsets are
  // local to the strip and clamped into it: at a strip edge that is not an image border the clamped reads feed only halo
  // columns, whose values no own column depends on)
  const bool act = tid < RW;
  const int j = act ? tid : 0, Xl = j / C, c = j - Xl * C, X = xs0 + Xl;
  auto loc = [&](int xg) { return min(max(xg - xs0, 0), SWh - 1) * C + c; };
  const int fxm = loc(refl(X - 1, W2)), fxp = loc(refl(X + 1, W2));                   // forward neighbours (REFLECT)
  const int txm = loc(max(X - 1, 0)), txp = loc(min(X + 1, W2 - 1));                  // adjoint neighbours, multiplicities
  const float mxm = X == 0 ? 0.f : (X == 1 ? 2.f : 1.f), mxp = X == W2 - 1 ? 0.f : (X == W2 - 2 ? 2.f : 1.f);
  int exl, exh; float etx;
  { int xlo, xhi; axis2x(X, W, xlo, xhi, etx); exl = (xlo - el0) * C + c; exh = (xhi - el0) * C + c; }
  const bool own_col = X >= 2 * c0 && X < 2 * c1;
  __syncthreads();

  // ---- up = resize2x(e) ----------------------------------------------------------------------------------------------------------
  if (act) {
    for (int p = Pu0; p < Pu1; ++p) {
      int ylo, yhi; float ty;
      axis2x(p, H, ylo, yhi, ty);
      const float* el = sE + (ylo - e0) * rw;
      const float* eh = sE + (yhi - e0) * rw;
      const float tl = el[exl], tr = el[exh], bl = eh[exl], br = eh[exh];
      const float top = tl + (tr - tl) * etx, bot = bl + (br - bl) * etx;
      bufA[(p - Pu0) * RW + j] = top + (bot - top) * ty;
    }
  }
  __syncthreads();
  // ---- base = G(1.2489996) up, REFLECT: per source row the pair (centre-row sum, edge-row sum) --------------------------------------
  if (act) {
    // blur = rows weighted (ba, bc, ba) of the horizontally blurred rows hb(r) = bc v0 + ba (vm + vp); r: reflected image row
    auto hb = [&](int r) { const float* q = bufA + (r - Pu0) * RW; return bc * q[j] + ba * (q[fxm] + q[fxp]); };
    float hp = hb(refl(Pa0 - 1, H2)), hm = hb(Pa0);
    for (int p = Pa0; p < Pa1; ++p) {
      const float hn = hb(refl(p + 1, H2));
      bufB[(p - Pa0) * RW + j] = bc * hm + ba * (hp + hn);
      hp = hm; hm = hn;
    }
  }
  __syncthreads();
  // ---- pyramid differences: loss and sign bytes --------------------------------------------------------------------------------------
  float lacc = 0.f;
  if (act) {
    // per source row: its own value and the sum of its two horizontal neighbours
    auto rowv = [&](int r, float& v0, float& vs) { const float* q = bufB + (r - Pa0) * RW; v0 = q[j]; vs = q[fxm] + q[fxp]; };
    float t0, ts, m0, ms;
    rowv(refl(Ps0 - 1, H2), t0, ts);
    rowv(Ps0, m0, ms);
    for (int p = Ps0; p < Ps1; ++p) {
      float n0, ns;
      rowv(refl(p + 1, H2), n0, ns);
      const float centre = m0, edge = ms + t0 + n0, corner = ts + ns;
      float bl[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) bl[k] = gc[k] * gc[k] * centre + ga[k] * gc[k] * edge + ga[k] * ga[k] * corner;
      const bool own = own_col && p >= 2 * r0 && p < 2 * r1;
      unsigned code = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = bl[k + 1] - bl[k];
        if (own) lacc += fabsf(d);
        code |= (d > 0.f ? 2u : (d < 0.f ? 0u : 1u)) << (2 * k);
      }
      sS[(p - Ps0) * RW + j] = (unsigned char)code;
      t0 = m0; ts = ms; m0 = n0; ms = ns;
    }
  }
  __syncthreads();
  // ---- d base = sum_k sign(d_k) (G(s_k+1) - G(s_k))^T / n * weight -------------------------------------------------------------------
  const float gpos = g.weight * inv_n;
  if (act) {
    // per source row r: what it contributes as the centre row (A) and as an edge row (Bv) of the adjoint stencil
    auto rowv = [&](int r, float& A, float& Bv) {
      const unsigned char* q = sS + (r - Ps0) * RW;
      const unsigned c0 = q[j], cm = q[txm], cp = q[txp];
      A = sLut[c0] + mxm * sLut[256 + cm] + mxp * sLut[256 + cp];
      Bv = sLut[256 + c0] + mxm * sLut[512 + cm] + mxp * sLut[512 + cp];
    };
    float Bp = 0.f, Am, Bm, dA;
    if (Pb0 > 0) rowv(Pb0 - 1, dA, Bp);
    rowv(Pb0, Am, Bm);
    for (int p = Pb0; p < Pb1; ++p) {
      float An = 0.f, Bn = 0.f;
      if (p + 1 < H2) rowv(p + 1, An, Bn);
      const float mym = p == 0 ? 0.f : (p == 1 ? 2.f : 1.f), myp = p == H2 - 1 ? 0.f : (p == H2 - 2 ? 2.f : 1.f);
      bufA[(p - Pb0) * RW + j] = gpos * (Am + mym * Bp + myp * Bn);
      Bp = Bm; Am = An; Bm = Bn;
    }
  }
  __syncthreads();
  // ---- d up = G(1.2489996)^T d base ----------------------------------------------------------------------------------------------------
  if (act) {
    auto rowv = [&](int r) { const float* q = bufA + (r - Pb0) * RW; return bc * q[j] + ba * (mxm * q[txm] + mxp * q[txp]); };
    float Rp = 0.f, Rm;
    if (Pd0 > 0) Rp = rowv(Pd0 - 1);
    Rm = rowv(Pd0);
    for (int p = Pd0; p < Pd1; ++p) {
      float Rn = 0.f;
      if (p + 1 < H2) Rn = rowv(p + 1);
      const float mym = p == 0 ? 0.f : (p == 1 ? 2.f : 1.f), myp = p == H2 - 1 ? 0.f : (p == H2 - 2 ? 2.f : 1.f);
      bufB[(p - Pd0) * RW + j] = bc * Rm + ba * (mym * Rp + myp * Rn);
      Rp = Rm; Rm = Rn;
    }
  }
  __syncthreads();
  // ---- d y += R^T d up  (up2x_bwd_kernel<1>, accumulate) --------------------------------------------------------------------------
  const int ow = (c1 - c0) * C;                            // the strip's own low-resolution columns
  for (int i = tid; i < (r1 - r0) * ow; i += NT) {
    const int row = i / ow, jj = i - row * ow, ix = c0 + jj / C, cc = jj % C, iy = r0 + row;
    float wy[4], wx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oy = 2 * iy - 1 + k, ox = 2 * ix - 1 + k;
      wy[k] = 0.f; wx[k] = 0.f;
      if (oy >= 0 && oy < H2) {
        int lo, hi; float tt; axis2x(oy, H, lo, hi, tt);
        wy[k] = (lo == iy ? 1.f - tt : 0.f) + (hi == iy ? tt : 0.f);
      }
      if (ox >= 0 && ox < W2) {
        int lo, hi; float tt; axis2x(ox, W, lo, hi, tt);
        wx[k] = (lo == ix ? 1.f - tt : 0.f) + (hi == ix ? tt : 0.f);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float w = wy[ky] * wx[kx];
        if (w != 0.f) s += w * bufB[(2 * iy - 1 + ky - Pd0) * RW + (2 * ix - 1 + kx - xs0) * C + cc];
      }
    float* o = g.dy + ((size_t)(b * H + iy) * W + ix) * C + cc;
    o[0] = o[0] + s;
  }
  // ---- loss ----------------------------------------------------------------------------------------------------------------------
  __shared__ float sblk[16];
  lacc = wave_sum(lacc);
  if ((tid & 63) == 0) sblk[tid >> 6] = lacc;
  __syncthreads();
  if (tid == 0 && g.loss) {
    float t2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t2 += sblk[k];
    atomicAdd(g.loss, t2 * inv_n);
  }
}

// ------------------------------------------------------------------------------------------------------------
// scalar losses with gradients
// ------------------------------------------------------------------------------------------------------------
// loss += wl * mean|a - b|;  da (+)= wg * sign(a-b)/n      (train.py:311-313, :325; b may be null)
__global__ void __launch_bounds__(256) l1_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                 float wl, float wg, float* loss, float* __restrict__ da,
                                                 int accumulate) {
  const float inv_n = 1.f / (float)n;
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = a[i] - (b ? b[i] : 0.f);
    acc += fabsf(d);
    if (da) {
      const float g = d > 0.f ? wg * inv_n : (d < 0.f ? -wg * inv_n : 0.f);
      da[i] = accumulate ? da[i] + g : g;
    }
  }
  block_atomic_add(acc, inv_n * wl, loss);
}

// the same on 16-byte groups (n % 4 == 0, 16-byte aligned tensors): one group per thread and trip, 32-bit indices.  The scalar
// kernel above spends ~10 us on a 1 M-element pooled feature map (8 dependent trips per thread behind 64-bit index arithmetic);
// three of them sit on each half of the perceptual term, which the backward pass waits for.
__global__ void __launch_bounds__(256) l1_vec4_kernel(const float4* __restrict__ a, const float4* __restrict__ b, unsigned n4,
                                                      float inv_n, float wl, float wg, float* loss, float4* __restrict__ da,
                                                      int accumulate) {
  float acc = 0.f;
  const float gp = wg * inv_n;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n4; i += gridDim.x * 256u) {
    const float4 x = a[i];
    const float4 y = b ? b[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float d[4] = {x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w};
    acc += (fabsf(d[0]) + fabsf(d[1])) + (fabsf(d[2]) + fabsf(d[3]));
    if (da) {
      float4 g = make_float4(d[0] > 0.f ? gp : (d[0] < 0.f ? -gp : 0.f), d[1] > 0.f ? gp : (d[1] < 0.f ? -gp : 0.f),
                             d[2] > 0.f ? gp : (d[2] < 0.f ? -gp : 0.f), d[3] > 0.f ? gp : (d[3] < 0.f ? -gp : 0.f));
      if (accumulate) { const float4 q = da[i]; g.x += q.x; g.y += q.y; g.z += q.z; g.w += q.w; }
      da[i] = g;
    }
  }
  block_atomic_add(acc, inv_n * wl, loss);
}

// LSGAN: loss += wl * mean((x - target)^2); dx = wg * 2 (x - target)/n     (train.py:234-237)
__global__ void __launch_bounds__(256) mse_kernel(const float* __restrict__ x, float target, size_t n, float wl, float wg,
                                                  float* loss, float* __restrict__ dx) {
  const float inv_n = 1.f / (float)n;
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = x[i] - target;
    acc += d * d;
    if (dx) dx[i] = wg * 2.f * d * inv_n;
  }
  block_atomic_add(acc, inv_n * wl, loss);
}

// Keras KLDivergence (train.py:232,305): loss += mean_b sum_j yt log(yt/yp), both clipped to [1e-7,1];
// dcmf = -yt/yp / B inside the clip range of yp, else 0
__global__ void __launch_bounds__(256) kl_kernel(const float* __restrict__ gt, const float* __restrict__ cmf, int B, int N,
                                                 float* loss, float* __restrict__ dcmf) {
  const size_t n = (size_t)B * N;
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float yt = fminf(fmaxf(gt[i], 1e-7f), 1.f);
    const float p = cmf[i];
    const float yp = fminf(fmaxf(p, 1e-7f), 1.f);
    acc += yt * logf(yt / yp);
    if (dcmf) dcmf[i] = (p > 1e-7f && p < 1.f) ? -yt / yp / (float)B : 0.f;
  }
  block_atomic_add(acc, 1.f / (float)B, loss);
}

// dz = cmf * (dcmf - sum_j dcmf_j cmf_j) * [z > 0]      (softmax + the ReLU in front of it; one block per row)
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const float* __restrict__ cmf, const float* __restrict__ dcmf,
                                                          const float* __restrict__ z, int N, float* __restrict__ dz) {
  __shared__ float sred[4];
  const size_t row = (size_t)blockIdx.x * N;
  float s = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) s += dcmf[row + n] * cmf[row + n];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
  __syncthreads();
  const float dot = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  for (int n = threadIdx.x; n < N; n += 256)
    dz[row + n] = z[row + n] > 0.f ? cmf[row + n] * (dcmf[row + n] - dot) : 0.f;
}

// the same with the row in registers: 1024 threads x RV 16-byte groups (N <= 4096 RV), both passes from one set of loads - a
// 4096-bin row on 256 threads was two passes of 16 dependent trips on 32 blocks: 14 us for 1.5 MB
template <int RV>
__global__ void __launch_bounds__(1024) softmax_bwd_vec_kernel(const float4* __restrict__ cmf, const float4* __restrict__ dcmf,
                                                               const float4* __restrict__ z, int n4, float4* __restrict__ dz) {
  __shared__ float sred[16];
  const size_t row = (size_t)blockIdx.x * n4;
  float4 c[RV], d[RV], zz[RV];
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < RV; ++r) {
    const int q = threadIdx.x + r * 1024;
    if (q < n4) {
      c[r] = cmf[row + q]; d[r] = dcmf[row + q]; zz[r] = z[row + q];
      s += (d[r].x * c[r].x + d[r].y * c[r].y) + (d[r].z * c[r].z + d[r].w * c[r].w);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = s;
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) dot += sred[w];
#pragma unroll
  for (int r = 0; r < RV; ++r) {
    const int q = threadIdx.x + r * 1024;
    if (q < n4)
      dz[row + q] = make_float4(zz[r].x > 0.f ? c[r].x * (d[r].x - dot) : 0.f, zz[r].y > 0.f ? c[r].y * (d[r].y - dot) : 0.f,
                                zz[r].z > 0.f ? c[r].z * (d[r].z - dot) : 0.f, zz[r].w > 0.f ? c[r].w * (d[r].w - dot) : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------------------
// blend / decoder tail / sun radiance backward
// ------------------------------------------------------------------------------------------------------------
// y_g = (1-a) sky + a sun,  y_lin = decomp(y_g).  Given dL/dy_g (dyg, nullable) and dL/dy_lin (dyl, nullable):
//   t = dyg + dyl * decomp'(y_g);  dsky = (1-a) t;  dsun = a t          (alpha is a constant: train.py:257)
__global__ void blend_bwd_kernel(const float* __restrict__ y_gamma, const float* __restrict__ alpha,
                                 const float* __restrict__ dyg, const float* __restrict__ dyl, size_t n,
                                 float* __restrict__ dsky, float* __restrict__ dsun) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float t = dyg ? dyg[i] : 0.f;
    if (dyl) t += dyl[i] * expf(y_gamma[i] * LN11) * (LN11 / 10.f);
    const float a = alpha[i];
    dsky[i] = (1.f - a) * t;
    dsun[i] = a * t;
  }
}

// The first stretch of the generator's backward pass in one launch (train.py:258-261,293-299 backwards; three channels per pixel):
//   t = dyg + (dyl + din[pixel][3 + c]) * d tone-map;  dsky = (1 - alpha) t, dsun = alpha t       (blend_bwd_kernel; din = the
//   adversarial term's gradient wrt the discriminator's 6-channel input, whose channels 3..5 are the prediction: slice + sum)
//   both decoder tails y = relu(res + lrelu(c, 0.1)) backwards                                      (decoder_tail_bwd_kernel x2)
// - five launches (slice_channels, axpby, blend_bwd, decoder_tail_bwd x2) on the dependent chain of the step, same arithmetic.
__global__ void __launch_bounds__(256) head_bwd_kernel(const float* __restrict__ y_gamma, const float* __restrict__ alpha,
                                                       const float* __restrict__ dyg, const float* __restrict__ dyl,
                                                       const float* __restrict__ din6, const float* __restrict__ y_f,
                                                       const float* __restrict__ res_f, const float* __restrict__ y_u,
                                                       const float* __restrict__ res_u, size_t n, float* __restrict__ dc_f,
                                                       float* __restrict__ dc_u, float* __restrict__ dres_u) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float t = dyg ? dyg[i] : 0.f;
    float l = dyl ? dyl[i] : 0.f;
    if (din6) { const size_t px = i / 3; l = l + din6[px * 6 + 3 + (i - px * 3)]; }
    if (dyl || din6) t += l * expf(y_gamma[i] * LN11) * (LN11 / 10.f);
    const float a = alpha[i];
    const float dsky = (1.f - a) * t, dsun = a * t;
    const float yf = y_f[i], yu = y_u[i];
    const float gf = yf > 0.f ? dsky : 0.f, gu = yu > 0.f ? dsun : 0.f;
    dc_f[i] = gf * ((yf - res_f[i]) > 0.f ? 1.f : 0.1f);
    dc_u[i] = gu * ((yu - res_u[i]) > 0.f ? 1.f : 0.1f);
    dres_u[i] = gu;
  }
}

// y = relu(res + lrelu(c, 0.1)):  g = dy*[y>0];  dres = g;  dc = g * (y - res > 0 ? 1 : 0.1)   (generator.py:119-124)
__global__ void decoder_tail_bwd_kernel(const float* __restrict__ y, const float* __restrict__ res,
                                        const float* __restrict__ dy, size_t n, float* __restrict__ dc,
                                        float* __restrict__ dres) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float yv = y[i];
    const float g = yv > 0.f ? dy[i] : 0.f;
    if (dres) dres[i] = g;
    dc[i] = g * ((yv - res[i]) > 0.f ? 1.f : 0.1f);
  }
}

// Backward of hdrsky_sun_rad.  One block per sample: d rad_gamma3 [B,P,3] ->
//   dpre[b][0..1] = d(pre-sigmoid gamma, beta),  dx[b][p] = dL/d(cmf/gmax),  dotx[b] = sum_p dx*cmf  (for the max term)
__global__ void __launch_bounds__(1024) sun_rad_bwd_kernel(const float* __restrict__ cmf, const unsigned int* gmax_bits,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ drg3, int P, float* __restrict__ dx,
                                                          float* __restrict__ dpre, float* __restrict__ dotx, int n,
                                                          int* __restrict__ claimed, int S, float* __restrict__ part) {
  // S == 1: block = sample (the 32x128 maps).  S > 1: block = (sample, pixel slice) - a 128x512 map is 65 536 pixels and one
  // block per sample was 8 blocks on 256 CUs (293 us per call) -, partial sums to part[b][s][3], summed in slice order by
  // sun_rad_bwd_fin_kernel
  // (256 or 1024 threads: a 4096-pixel slice on 256 threads was 16 dependent trips per thread on 32 blocks, 20 us)
  __shared__ float sred[3][16];
  const int b = blockIdx.x / S, sl = blockIdx.x % S;
  const int per = (P + S - 1) / S;
  const int p0 = sl * per, p1 = min(P, p0 + per);
  const int NTH = blockDim.x, nw = NTH >> 6;
  const float gmax = __uint_as_float(*gmax_bits);
  const float g = gamma[b], bt = beta[b];
  const float D = bt * 1.7724539f + 1e-5f, be = bt + 1e-5f;
  float sg = 0.f, sb = 0.f, sd = 0.f;
  for (int p = p0 + threadIdx.x; p < p1; p += NTH) {
    const size_t i = (size_t)b * P + p;
    const float x = cmf[i] / gmax;
    const float d1 = 1.f - x;
    const float E = expf(-(d1 * d1) / be);
    const float r = g * E / D;
    float dr = 0.f;
    if (!(r > 30000.f)) {
      const float drg = drg3[i * 3] + drg3[i * 3 + 1] + drg3[i * 3 + 2];
      dr = drg * 10.f / ((1.f + 10.f * r) * LN11);
    }
    sg += dr * E / D;
    sb += dr * (g * E * (d1 * d1) / (be * be) / D - g * E * 1.7724539f / (D * D));
    const float dxi = dr * g * E * 2.f * d1 / be / D;
    dx[i] = dxi;
    sd += dxi * cmf[i];
    // every element of the batch tensor that equals its maximum receives an equal share of the maximum's gradient
    // (second launch) - tf.reduce_max's gradient (indicators / num_selected) - so the tied elements are counted here: an
    // integer count, the same whatever order the blocks arrive in.  (An atomicCAS "first to arrive" claim gave the whole
    // term to any one of several tied elements - the gradients of the sun-pose net then repeated only to round-off.)
    if (cmf[i] == gmax) atomicAdd(claimed, 1);
  }
  sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd);
  if ((threadIdx.x & 63) == 0) { sred[0][threadIdx.x >> 6] = sg; sred[1][threadIdx.x >> 6] = sb; sred[2][threadIdx.x >> 6] = sd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tg = 0.f, tb = 0.f, td = 0.f;
    for (int w = 0; w < nw; ++w) { tg += sred[0][w]; tb += sred[1][w]; td += sred[2][w]; }
    if (S == 1) {
      dpre[b * 2 + 0] = tg * g * (1.f - g);
      dpre[b * 2 + 1] = tb * bt * (1.f - bt);
      dotx[b] = td;
    } else {
      float* q = part + ((size_t)b * S + sl) * 3;
      q[0] = tg; q[1] = tb; q[2] = td;
    }
  }
}

__global__ void __launch_bounds__(64) sun_rad_bwd_fin_kernel(const float* __restrict__ part, int S, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ dpre,
                                                             float* __restrict__ dotx) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  float tg = 0.f, tb = 0.f, td = 0.f;
  for (int s = 0; s < S; ++s) { const float* q = part + ((size_t)b * S + s) * 3; tg += q[0]; tb += q[1]; td += q[2]; }
  const float g = gamma[b], bt = beta[b];
  dpre[b * 2 + 0] = tg * g * (1.f - g);
  dpre[b * 2 + 1] = tb * bt * (1.f - bt);
  dotx[b] = td;
}

// dcmf (+)= dx/gmax, and at the arg-max element(s) of the whole batch: -= sum_b dotx[b] / gmax^2 / (number of them)
// (tf.reduce_max over the batch tensor, generator.py:160; ties share the gradient evenly as in TF's _MinOrMaxGrad)
// rec: nrec records of (dotx[B], tie count) - one for a batch on one GPU; one per replica (all-gathered) when the maximum
// runs over the batch of every replica (parallel.py sync_batch_stats)
__global__ void sun_rad_bwd_cmf_kernel(const float* __restrict__ cmf, const unsigned int* gmax_bits,
                                       const float* __restrict__ dx, const float* __restrict__ rec, int nrec, int B, int P,
                                       float* __restrict__ dcmf) {
  const float gmax = __uint_as_float(*gmax_bits);
  float tot = 0.f;
  int ties = 0;
  for (int r = 0; r < nrec; ++r) {
    const float* dotx = rec + (size_t)r * (B + 1);
    for (int b = 0; b < B; ++b) tot += dotx[b];
    ties += __float_as_int(dotx[B]);
  }
  const size_t n = (size_t)B * P;
  const float share = ties > 0 ? tot / (gmax * gmax) / (float)ties : 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float g = dx[i] / gmax;
    if (cmf[i] == gmax) g -= share;
    dcmf[i] += g;
  }
}

// d flat[b][i] = dpre[b][0]*kg[i] + dpre[b][1]*kb[i];  dkg[i] += sum_b dpre[b][0]*flat[b][i] (flat recomputed from x);
// dbias handled by the caller (sum_b dpre).   (sunrad_net.py:52-53)
__global__ void dense_heads_bwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                       const float* __restrict__ shift, float slope, int B, int F, int C,
                                       const float* __restrict__ kg, const float* __restrict__ kb,
                                       const float* __restrict__ dpre, float* __restrict__ dact, float* __restrict__ dkg,
                                       float* __restrict__ dkb, float* dbg, float* dbb) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F; i += gridDim.x * blockDim.x) {
    const int c = i % C;
    const float wg = kg[i], wb = kb[i];
    float ag = 0.f, ab = 0.f;
    const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
    for (int b0 = 0; b0 < B; b0 += 8) {       // eight samples' loads in flight (one per trip was a chain of B round trips: 33 us)
      float v8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v8[k] = x[(size_t)min(b0 + k, B - 1) * F + i];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int b = b0 + k;
        if (b < B) {
          const float f = leaky(scale ? v8[k] * sc + sh : v8[k], slope);
          ag += dpre[b * 2] * f; ab += dpre[b * 2 + 1] * f;
          dact[(size_t)b * F + i] = dpre[b * 2] * wg + dpre[b * 2 + 1] * wb;
        }
      }
    }
    dkg[i] += ag; dkb[i] += ab;
    if (i == 0) {
      float sg = 0.f, sb = 0.f;
      for (int b = 0; b < B; ++b) { sg += dpre[b * 2]; sb += dpre[b * 2 + 1]; }
      dbg[0] += sg; dbb[0] += sb;
    }
  }
}

// out[b,p,0..c_take) = x[b,p,c_off..c_off+c_take) * scale      (d y_lin = channels 3..5 of the disc input gradient)
__global__ void slice_channels_kernel(const float* __restrict__ x, size_t npix, int C, int c_off, int c_take, float scale,
                                      int accumulate, float* __restrict__ out) {
  const size_t n = npix * c_take;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[(i / c_take) * C + c_off + (i % c_take)] * scale;
    out[i] = accumulate ? out[i] + v : v;
  }
}

// out[p, 0..Cpad) = [x[p, 0..C), zeros]: channel padding of an image / of a filter's input-channel axis (the
// distortion-aware kernels read 32-channel groups: the 3-channel input layer of the sun-pose net runs on a padded copy)
__global__ void pad_channels_kernel(const float* __restrict__ x, size_t npix, int C, int Cpad, float* __restrict__ out) {
  const size_t n = npix * Cpad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    out[i] = c < C ? x[(i / Cpad) * C + c] : 0.f;
  }
}

// concat along channels: out[b,p,:] = [a (Ca), b (Cb)]      (discriminator.py:43)
__global__ void concat2_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb, size_t npix,
                               float* __restrict__ out) {
  const int C = Ca + Cb;
  const size_t n = npix * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t p = i / C;
    out[i] = c < Ca ? a[p * Ca + c] : b[p * Cb + (c - Ca)];
  }
}


// y = bf16(leaky(x * scale + shift, slope)) with the per-(sample, channel) affine of the consumer-side operand transform
// (hdrsky_conv_desc in_mode: none / affine table / InstanceNorm finalised from the producer's partials, the formula of the
// conv and weight-gradient staging): the activated input x' of a conv as a final bf16 tensor, what the LDS-DMA weight-
// gradient kernel (conv_wgrad2_kernel) reads.  Block = (sample, slice of its pixels); thread = (pixel, 8 channels).
__global__ void __launch_bounds__(256) act_bf16_kernel(const float* __restrict__ x, int HW, int C, int mode,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int ss_bstride, const float* __restrict__ part, int nparts,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       float slope, uint4* __restrict__ y, int bps, int x16) {
  __shared__ float sSc[1024], sSh[1024];
  __shared__ float sP[4096];
  const int b = blockIdx.x / bps, blk = blockIdx.x % bps;
  const int nq = C >> 3, nitems = HW * nq;
  const int per = (nitems + bps - 1) / bps;
  const int i1 = min(nitems, (blk + 1) * per);
  const size_t xo = (size_t)b * HW * C;
  const bool xf = mode != HDRSKY_IN_NONE;
  constexpr int UNR = 4;                                   // items per thread in flight (all loads before the first use)
  float4 va[UNR], vb[UNR];
  int i0 = blk * per + threadIdx.x;
  if (i0 < i1) {                                           // the first (normally only) round's loads fly under the table prologue
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = min(i0 + u * 256, i1 - 1);
      ld8any(x, x16, xo + (size_t)i * 8, va[u], vb[u]);
    }
  }
  if (mode == HDRSKY_IN_PARTIALS && C <= 256) {
    // the sample's tile partials ([tile][sum | squares][C], contiguous) go through LDS a chunk at a time - every thread
    // copies, all loads in flight - and thread c adds its channel's entries in tile order: the same additions as
    // in_partial_sums (bit-identical statistics), without a chain of nparts / 8 dependent global round trips in front of
    // every block (64 tiles at 32x128: ~5 us, more than the block's own work)
    const int pch = 2048 / C;
    const float* pb = part + (size_t)b * nparts * 2 * C;
    float s0 = 0.f, ss = 0.f;
    for (int p0 = 0; p0 < nparts; p0 += pch) {
      const int np = min(pch, nparts - p0), n = np * 2 * C;
      for (int i = threadIdx.x; i < n; i += 256) sP[i] = pb[(size_t)p0 * 2 * C + i];
      __syncthreads();
      if ((int)threadIdx.x < C)
        for (int k = 0; k < np; ++k) { s0 += sP[k * 2 * C + threadIdx.x]; ss += sP[k * 2 * C + C + threadIdx.x]; }
      __syncthreads();
    }
    if ((int)threadIdx.x < C) {
      const int c = threadIdx.x;
      const float inv_count = 1.f / (float)HW;
      const float mean = s0 * inv_count;
      const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
      const float sc = gamma[c] / sqrtf(var + eps);
      sSc[c] = sc; sSh[c] = beta[c] - mean * sc;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      float sc = 1.f, sh = 0.f;
      if (mode == HDRSKY_IN_AFFINE) {
        sc = scale[b * ss_bstride + c]; sh = shift[b * ss_bstride + c];
      } else if (mode == HDRSKY_IN_PARTIALS) {
        float s0, ss;
        in_partial_sums(part + (size_t)b * nparts * 2 * C + c, nparts, C, s0, ss);
        const float inv_count = 1.f / (float)HW;
        const float mean = s0 * inv_count;
        const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
        sc = gamma[c] / sqrtf(var + eps);
        sh = beta[c] - mean * sc;
      }
      sSc[c] = sc; sSh[c] = sh;
    }
  }
  __syncthreads();
  for (bool first = true; i0 < i1; i0 += 256 * UNR, first = false) {
    if (!first) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int i = min(i0 + u * 256, i1 - 1);
        ld8any(x, x16, xo + (size_t)i * 8, va[u], vb[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = i0 + u * 256;
      if (i >= i1) break;
      const int qc = i % nq;
      const float in[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = xf ? leaky(in[j] * sSc[qc * 8 + j] + sSh[qc * 8 + j], slope) : leaky(in[j], slope);
      uint4 hi, lo;
      pack8<false>(v, hi, lo);
      y[(size_t)b * nitems + i] = hi;
    }
  }
}

// out[m, :] = [s0[m, :w0] | s1[m, :w1] | s2[m, :w2] | s3[m, :w3]]: the four Dense operands of a replica (flat | df1 | f1 | dz)
// as ONE row block for the all-gather of the gather_dense exchange (parallel.py) - 16-byte copies, widths % 4 == 0
struct ConcatRows4 { const float4* src[4]; int w4[4]; };
__global__ void concat_rows4_kernel(ConcatRows4 j, int M, float4* __restrict__ out) {
  const int W4 = j.w4[0] + j.w4[1] + j.w4[2] + j.w4[3];
  const size_t n = (size_t)M * W4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / W4;
    int c = (int)(i - m * W4), k = 0;
    while (c >= j.w4[k]) { c -= j.w4[k]; ++k; }
    out[i] = j.src[k][m * j.w4[k] + c];
  }
}

// VGG input: x*255 - mean[c]  (vgg16.py:133-141); backward is a multiply by 255 (folded into the caller's scale)
__global__ void vgg_pre_kernel(const float* __restrict__ x, size_t n, float* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    y[i] = x[i] * 255.f - (c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f));
  }
}

// y = a*sa + b*sb  (b nullable)
__global__ void axpby_kernel(const float* __restrict__ a, float sa, const float* __restrict__ b, float sb, size_t n,
                             float* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = a[i] * sa + (b ? b[i] * sb : 0.f);
}

// ------------------------------------------------------------------------------------------------------------
// Dense weight / bias gradient: dW[K][N] += x[M][K]^T dy[M][N], db[N] += sum_m dy   (M <= 32: HBM-write bound)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fc_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int M,
                                                       int K, int N, int accumulate, float* __restrict__ dw,
                                                       float* __restrict__ db) {
  const int n4 = N >> 2;
  const int k0 = blockIdx.y * 8;
  const int nq = blockIdx.x * 256 + threadIdx.x;
  if (nq >= n4) return;
  float4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m = 0; m < M; ++m) {
    const float4 d = reinterpret_cast<const float4*>(dy + (size_t)m * N)[nq];
    bs.x += d.x; bs.y += d.y; bs.z += d.z; bs.w += d.w;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xv = x[(size_t)m * K + k0 + k];   // block-uniform address: a scalar load
      acc[k].x += xv * d.x; acc[k].y += xv * d.y; acc[k].z += xv * d.z; acc[k].w += xv * d.w;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float4* dst = reinterpret_cast<float4*>(dw + (size_t)(k0 + k) * N) + nq;
    float4 o = accumulate ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x += acc[k].x; o.y += acc[k].y; o.z += acc[k].z; o.w += acc[k].w;
    *dst = o;
  }
  if (db && blockIdx.y == 0) {
    float4* dst = reinterpret_cast<float4*>(db) + nq;
    float4 o = accumulate ? *dst : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x += bs.x; o.y += bs.y; o.z += bs.z; o.w += bs.w;
    *dst = o;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Keras-2 OptimizerV2 Adam (train_sun.py:191 via tf_utils.py:324; beta1 0.9, beta2 0.999, eps 1e-7) over one flat
// parameter buffer:  m <- b1*m + (1-b1)*g ; v <- b2*v + (1-b2)*g^2 ; w <- w - lr_t * m / (sqrt(v) + eps)
// with lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) computed by the caller (t = step count).
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n4, float lr_t, float b1, float b2,
                                                   float eps, float gscale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 wv = reinterpret_cast<float4*>(w)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float* wp = &wv.x; float* mp = &mv.x; float* vp = &vv.x; const float* gp = &gv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * gscale;
      mp[k] = b1 * mp[k] + (1.f - b1) * gg;
      vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
      wp[k] -= lr_t * mp[k] / (sqrtf(vp[k]) + eps);
    }
    reinterpret_cast<float4*>(w)[i] = wv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Keras-2 OptimizerV2 RMSprop (train.py:201-202; rho 0.9, momentum 0, eps 1e-7 OUTSIDE the sqrt) over one flat
// parameter buffer:  ms <- rho*ms + (1-rho)*g^2 ;  w <- w - lr*g/(sqrt(ms)+eps).  gscale averages replica sums.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rmsprop_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                      float* __restrict__ ms, size_t n4, float lr, float rho, float eps,
                                                      float gscale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 wv = reinterpret_cast<float4*>(w)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(ms)[i];
    const float gs[4] = {gv.x * gscale, gv.y * gscale, gv.z * gscale, gv.w * gscale};
    float ws[4] = {wv.x, wv.y, wv.z, wv.w}, m[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = rho * m[j] + (1.f - rho) * gs[j] * gs[j];
      ws[j] -= lr * gs[j] / (sqrtf(m[j]) + eps);
    }
    reinterpret_cast<float4*>(w)[i] = make_float4(ws[0], ws[1], ws[2], ws[3]);
    reinterpret_cast<float4*>(ms)[i] = make_float4(m[0], m[1], m[2], m[3]);
  }
}

// the same over TWO flat buffers in one launch (the generator / sun-pose conv parameters and the discriminator's: two launches at
// the very end of the step, where nothing else is left to overlap their latency): blocks [0, nb1) walk buffer 1, the rest buffer 2
__global__ void __launch_bounds__(256) rmsprop2_kernel(float* __restrict__ w1, const float* __restrict__ g1, float* __restrict__ ms1,
                                                       size_t n41, int nb1, float* __restrict__ w2, const float* __restrict__ g2,
                                                       float* __restrict__ ms2, size_t n42, float lr, float rho, float eps, float gscale,
                                                       int nt) {
  typedef __attribute__((ext_vector_type(4))) float f32v4_t;
  const bool second = (int)blockIdx.x >= nb1;
  float* w = second ? w2 : w1; const float* g = second ? g2 : g1; float* ms = second ? ms2 : ms1;
  const size_t n4 = second ? n42 : n41;
  const size_t nblk = second ? gridDim.x - nb1 : nb1, blk = second ? blockIdx.x - nb1 : blockIdx.x;
  for (size_t i = blk * (size_t)blockDim.x + threadIdx.x; i < n4; i += nblk * blockDim.x) {
    float4 wv = reinterpret_cast<float4*>(w)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(ms)[i];
    const float gs[4] = {gv.x * gscale, gv.y * gscale, gv.z * gscale, gv.w * gscale};
    float ws[4] = {wv.x, wv.y, wv.z, wv.w}, m[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = rho * m[j] + (1.f - rho) * gs[j] * gs[j];
      ws[j] -= lr * gs[j] / (sqrtf(m[j]) + eps);
    }
    if (nt) {      // (tuning hook HDRSKY_OPT_NT: the updated weights / slots are next read one step later)
      __builtin_nontemporal_store(f32v4_t{ws[0], ws[1], ws[2], ws[3]}, reinterpret_cast<f32v4_t*>(w) + i);
      __builtin_nontemporal_store(f32v4_t{m[0], m[1], m[2], m[3]}, reinterpret_cast<f32v4_t*>(ms) + i);
    } else {
      reinterpret_cast<float4*>(w)[i] = make_float4(ws[0], ws[1], ws[2], ws[3]);
      reinterpret_cast<float4*>(ms)[i] = make_float4(m[0], m[1], m[2], m[3]);
    }
  }
}

// RMSprop of a Dense kernel [K][N] fused with the refresh of its two bf16 MFMA images (packed [K/8][N][8] for the
// forward, natural [K][N] for the data gradient): the weights are read and written once instead of three times.
// Thread = (8 consecutive k, one n): every access is coalesced across n and the packed row is one 16-byte store.
__global__ void __launch_bounds__(256) rmsprop_fc_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                         float* __restrict__ ms, int K, int N, float lr, float rho,
                                                         float eps, float gscale, uint4* __restrict__ pk_hi,
                                                         unsigned short* __restrict__ nat_hi) {
  const size_t total = (size_t)(K >> 3) * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % N);
    const size_t k0 = (i / N) << 3;
    unsigned short h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t idx = (k0 + j) * N + n;
      const float gg = g[idx] * gscale;
      const float m = rho * ms[idx] + (1.f - rho) * gg * gg;
      const float wv = w[idx] - lr * gg / (sqrtf(m) + eps);
      ms[idx] = m; w[idx] = wv;
      h[j] = f2bf(wv);
      if (nat_hi) __builtin_nontemporal_store(h[j], nat_hi + idx);      // (the images are written once and read by the NEXT step: round 5,
    }                                                                    //  non-temporal stores: profiles/r05_fc_nt.txt)
    const uint4 pk = uint4{h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16),
                           h[6] | ((unsigned)h[7] << 16)};
    __builtin_nontemporal_store(pk.x, &pk_hi[i].x); __builtin_nontemporal_store(pk.y, &pk_hi[i].y);
    __builtin_nontemporal_store(pk.z, &pk_hi[i].z); __builtin_nontemporal_store(pk.w, &pk_hi[i].w);
  }
}

// words [0, n) <- 0; head / tail by single words, the aligned middle by 16-byte stores
__global__ void __launch_bounds__(256) zero_words_kernel(unsigned* __restrict__ p, size_t n) {
  const size_t head = min(n, (size_t)((4 - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3)) & 3));
  const size_t n4 = (n - head) / 4, tail0 = head + n4 * 4;
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  for (size_t i = tid; i < n4; i += nth) q[i] = uint4{0u, 0u, 0u, 0u};
  if (tid < head) p[tid] = 0u;
  if (tid < n - tail0) p[tail0 + tid] = 0u;
}

__global__ void __launch_bounds__(256) zero_bytes_kernel(unsigned char* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0;
}

inline unsigned grid_for(size_t n, int per = 256) {
  size_t g = (n + per - 1) / per;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" {

int hdrsky_bn_train_finalize(const float* part, int nparts_total, int C, int count, const float* gamma, const float* beta,
                             float eps, float momentum, float* moving_mean, float* moving_var, float* mean, float* rstd,
                             float* scale, float* shift, int rows, void* stream) {
  if (!part || !gamma || !beta || !mean || !rstd || !scale || !shift || rows < 1) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(C), dim3(64), 0, S_(stream), part, nparts_total, C,
                     (float)count, gamma, beta, eps, momentum, moving_mean, moving_var, mean, rstd, scale, shift, rows);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

/* workspace: (2*nblocks*C + 2*C) floats, nblocks = hdrsky_bn_bwd_nblocks() */
int hdrsky_bn_bwd_nblocks(void) { return 128; }

int hdrsky_bn_act_bwd_reduce(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, float slope, int npix, int C, float* part, void* stream) {
  if (!x || !dy || !mean || !rstd || !gamma || !beta || !part || (C & 3) || C > 1024) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(128), dim3(256), 256 * 8 * sizeof(float), S_(stream), x, dy, mean, rstd,
                     gamma, beta, slope, (size_t)npix, C, part, 0);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_bn_act_bwd_apply(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, float slope, int npix, int C, const float* part_all, int nblocks_all,
                            double count_all, const float* part_local, int nblocks_local, float* m1m2, float* dgamma,
                            float* dbeta, void* dx, int dx_bf16, void* stream) {
  if (!x || !dy || !mean || !rstd || !gamma || !beta || !part_all || !part_local || !m1m2 || !dx || (C & 3) || C > 1024 ||
      nblocks_all < 1 || nblocks_local < 0 || !(count_all > 0.0))
    return HDRSKY_EINVAL;
  hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3(C), dim3(64), 0, S_(stream), part_all, nblocks_all, (float)count_all,
                     part_local, nblocks_local, C, m1m2, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((size_t)npix * C / 4)), dim3(256), 0, S_(stream), x, dy, mean,
                     rstd, gamma, beta, slope, m1m2, (size_t)npix * C / 4, C, dx, dx_bf16 & 1);   // (fp32 dy: _reduce has no flag)
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_bn_act_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, float slope, int npix, int C, float* workspace, float* dgamma, float* dbeta,
                      void* dx, int dx_bf16, void* stream) {
  if (!x || !dy || !mean || !rstd || !gamma || !beta || !workspace || !dx || (C & 3) || C > 1024) return HDRSKY_EINVAL;
  const int nb = 128;
  float* part = workspace;
  float* m1m2 = workspace + (size_t)2 * nb * C;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), 256 * 8 * sizeof(float), S_(stream), x, dy, mean, rstd,
                     gamma, beta, slope, (size_t)npix, C, part, dx_bf16 & 6);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, S_(stream), part, nb, C, (float)npix, m1m2,
                     dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((size_t)npix * C / 4)), dim3(256), 0, S_(stream), x, dy, mean,
                     rstd, gamma, beta, slope, m1m2, (size_t)npix * C / 4, C, dx, dx_bf16);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_affine_act_bwd(const float* x, const float* dy, const float* scale, const float* shift, float slope, size_t n,
                          int C, void* dx, int dx_bf16, void* stream) {
  if (!x || !dy || !dx) return HDRSKY_EINVAL;
  const bool vec = (C & 3) == 0 && (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) |
                                                    reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
  if (dx_bf16 && !vec) return HDRSKY_EUNSUPPORTED;      // (either bit: bf16 dx or bf16 dy need the 16-byte path)
  if (vec) hipLaunchKernelGGL(affine_act_bwd_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, S_(stream), x, dy, scale, shift, slope, n, C, dx, dx_bf16);
  else hipLaunchKernelGGL(affine_act_bwd_kernel<1>, dim3(grid_for(n)), dim3(256), 0, S_(stream), x, dy, scale, shift, slope, n, C, dx, 0);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_maxpool_fwd(const float* y, int B, int H, int W, int C, float* p, void* stream) {
  if (!y || !p || (C & 3) || ((H | W) & 1)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((size_t)B * H * W * C / 16)), dim3(256), 0, S_(stream), y, B, H, W, C, p);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_maxpool_relu_bwd(const float* y, const float* dp, int B, int H, int W, int C, float* dy, void* stream) {
  if (!y || !dp || !dy || (C & 3) || ((H | W) & 1)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(maxpool_relu_bwd_kernel, dim3(grid_for((size_t)B * H * W * C / 16)), dim3(256), 0, S_(stream), y, dp, B,
                     H, W, C, dy);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_maxpool_fwd_bf16(const void* y, int B, int H, int W, int C, float* p32, void* p16, void* stream) {
  if (!y || (!p32 && !p16) || (C & 3) || ((H | W) & 1)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(maxpool_fwd_bf16_kernel, dim3(grid_for((size_t)B * H * W * C / 16)), dim3(256), 0, S_(stream),
                     (const unsigned short*)y, B, H, W, C, p32, (unsigned short*)p16);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_maxpool_relu_bwd_bf16(const void* y, const float* dp, int B, int H, int W, int C, void* dy, int dy_bf16,
                                 void* stream) {
  if (!y || !dp || !dy || (C & 3) || ((H | W) & 1)) return HDRSKY_EINVAL;
  if (dy_bf16)
    hipLaunchKernelGGL((maxpool_relu_bwd_bf16_kernel<true, false>), dim3(grid_for((size_t)B * H * W * C / 16)), dim3(256), 0, S_(stream),
                       (const unsigned short*)y, dp, B, H, W, C, dy, nullptr, nullptr, 0.f, 0.f, nullptr);
  else
    hipLaunchKernelGGL((maxpool_relu_bwd_bf16_kernel<false, false>), dim3(grid_for((size_t)B * H * W * C / 16)), dim3(256), 0, S_(stream),
                       (const unsigned short*)y, dp, B, H, W, C, dy, nullptr, nullptr, 0.f, 0.f, nullptr);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

// the same with the L1 term of the block's pooled features folded in: dp' = (dp or 0) + wg * sign(pool - target) / n,
// *loss += wl * mean|pool - target|  (n = elements of pool)
int hdrsky_maxpool_relu_l1_bwd_bf16(const void* y, const float* pool, const float* target, const float* dp, int B, int H, int W, int C,
                                    float wl, float wg, float* loss, void* dy, int dy_bf16, void* stream) {
  if (!y || !pool || !target || !dy || (C & 3) || ((H | W) & 1)) return HDRSKY_EINVAL;
  const float inv_n = 1.f / ((float)B * (H / 2) * (W / 2) * C);
  if (dy_bf16)
    hipLaunchKernelGGL((maxpool_relu_bwd_bf16_kernel<true, true>), dim3(grid_for((size_t)B * H * W * C / 16)), dim3(256), 0, S_(stream),
                       (const unsigned short*)y, dp, B, H, W, C, dy, pool, target, wg * inv_n, wl * inv_n, loss);
  else
    hipLaunchKernelGGL((maxpool_relu_bwd_bf16_kernel<false, true>), dim3(grid_for((size_t)B * H * W * C / 16)), dim3(256), 0, S_(stream),
                       (const unsigned short*)y, dp, B, H, W, C, dy, pool, target, wg * inv_n, wl * inv_n, loss);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_act_bwd_bf16(const void* y, const float* dy, float slope, size_t n, void* dx, int dx_bf16, void* stream) {
  if (!y || !dy || !dx || (n & 3)) return HDRSKY_EINVAL;
  if (n == 0) return HDRSKY_OK;
  hipLaunchKernelGGL(act_bwd_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, S_(stream), (const unsigned short*)y, dy,
                     slope, n / 4, dx, dx_bf16);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_up2x_fwd(const float* a, const float* b, int B, int H, int W, int C, float* y, void* stream) {
  if (!a || !y) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(up2x_fwd_kernel, dim3(grid_for((size_t)B * 4 * H * W * C)), dim3(256), 0, S_(stream), a, b, B, H, W, C, y);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_up2x_xf_bf16(const float* x, int x_bf16, int B, int H, int W, int C, const float* in_part, int in_nparts, const float* gamma,
                        const float* beta, float eps, float slope, void* y_bf16, void* stream) {
  if (!x || !y_bf16 || (C & 7) || C > 512 || B <= 0) return HDRSKY_EINVAL;
  if (in_part && (!gamma || !beta || in_nparts <= 0)) return HDRSKY_EINVAL;
  const int nitems = 4 * H * W * (C / 8);
  int bps = cdiv(nitems, 256 * 8);          // ~8 items per thread
  if (bps < 1) bps = 1;
  hipLaunchKernelGGL(up2x_xf_bf16_kernel, dim3(B * bps), dim3(256), 0, S_(stream), x, B, H, W, C, in_part, in_nparts, gamma,
                     beta, eps, slope, (uint4*)y_bf16, bps, x_bf16 ? 1 : 0);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

// hdrsky_up2x_xf_bf16 on a PAIRED tensor (samples [0, B/2): gamma / beta, samples [B/2, B): gamma2 / beta2)
int hdrsky_up2x_xf_bf16_pair(const float* x, int x_bf16, int B, int H, int W, int C, const float* in_part, int in_nparts, const float* gamma,
                             const float* beta, const float* gamma2, const float* beta2, float eps, float slope, void* y_bf16, void* stream) {
  if (!x || !y_bf16 || (C & 7) || C > 512 || B <= 0 || (B & 1) || !in_part || !gamma || !beta || !gamma2 || !beta2 || in_nparts <= 0) return HDRSKY_EINVAL;
  const int nitems = 4 * H * W * (C / 8);
  int bps = cdiv(nitems, 256 * 8);
  if (bps < 1) bps = 1;
  hipLaunchKernelGGL(up2x_xf_bf16_kernel, dim3(B * bps), dim3(256), 0, S_(stream), x, B, H, W, C, in_part, in_nparts, gamma,
                     beta, eps, slope, (uint4*)y_bf16, bps, x_bf16 ? 1 : 0, gamma2, beta2, B / 2);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_up2x_bwd(const float* dy, int B, int H, int W, int C, float scale, int accumulate, float* dx, void* stream) {
  if (!dy || !dx) return HDRSKY_EINVAL;
  if ((C & 3) == 0)
    hipLaunchKernelGGL(up2x_bwd_kernel<4>, dim3(grid_for((size_t)B * H * W * C / 4)), dim3(256), 0, S_(stream), dy, B, H, W, C,
                       scale, accumulate, dx);
  else
    hipLaunchKernelGGL(up2x_bwd_kernel<1>, dim3(grid_for((size_t)B * H * W * C)), dim3(256), 0, S_(stream), dy, B, H, W, C, scale,
                       accumulate, dx);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_blur3(const float* x, int B, int H, int W, int C, float sigma, int transpose, float* y, void* stream) {
  if (!x || !y || H < 4 || W < 4) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(blur3_kernel, dim3(grid_for((size_t)B * H * W * C)), dim3(256), 0, S_(stream), x, B, H, W, C, sigma,
                     transpose, y);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_dog_mid(const float* base, int B, int H, int W, int C, float weight, float* h, float* loss, void* stream) {
  if (!base || !h) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(dog_mid_kernel, dim3(grid_for((size_t)B * H * W * C) > 1024 ? 1024 : grid_for((size_t)B * H * W * C)), dim3(256), 0, S_(stream), base, B, H, W, C,
                     weight, h, loss);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_dog_mid_bwd(const float* h, int B, int H, int W, int C, float* dbase, void* stream) {
  if (!h || !dbase || H < 4 || W < 4) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(dog_mid_bwd_kernel, dim3(grid_for((size_t)B * H * W * C)), dim3(256), 0, S_(stream), h, B, H, W, C, dbase);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_dog_loss(const float* y, const float* t, int B, int H, int W, int C, float weight, float* loss, float* dy, void* stream) {
  if (!y || !t || !dy || B <= 0 || H < 4 || W < 4 || C <= 0) return HDRSKY_EINVAL;
  // a band of RB low-resolution rows needs (2 RB + 10) + (2 RB + 8) high-resolution rows of fp32, RB + 7 low-resolution ones
  // and 2 RB + 6 rows of sign bytes in LDS
  // column strips: a thread per (high-resolution column, channel) of the strip incl. its +-5 halo, at most 1024 of them
  if (C > 64) return HDRSKY_EUNSUPPORTED;
  int nstrips = 1, SW = W;
  if (2 * W * C > 1024) {
    const int swmax = (1024 / C - 10) / 2;                  // own low-resolution columns that fit beside the halo
    if (swmax < 8) return HDRSKY_EUNSUPPORTED;
    nstrips = cdiv(W, swmax);
    SW = cdiv(W, nstrips);
  }
  const int RW = (nstrips == 1 ? 2 * W : 2 * SW + 10) * C, rw = (nstrips == 1 ? W : SW + 7) * C;
  for (int RB = 4; RB >= 1; RB >>= 1) {
    DogArgs g{};
    g.SW = SW; g.nstrips = nstrips;
    g.offB = (2 * RB + 10) * RW * 4;
    g.offE = g.offB + (2 * RB + 8) * RW * 4;
    g.offL = g.offE + (RB + 7) * rw * 4;
    g.offS = g.offL + 3 * 256 * 4;
    const int lds = roundup(g.offS + (2 * RB + 6) * RW, 16);
    if (lds > 160 * 1024 - 256) continue;                 // (+ the kernel's static words)
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(dog_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) !=
          hipSuccess)
        return HDRSKY_ELAUNCH;
      attr_set = true;
    }
    g.y = y; g.t = t; g.loss = loss; g.dy = dy; g.B = B; g.H = H; g.W = W; g.C = C; g.RB = RB; g.nbands = cdiv(H, RB);
    g.weight = weight;
    hipLaunchKernelGGL(dog_fused_kernel, dim3(B * g.nbands * g.nstrips), dim3(1024), lds, S_(stream), g);
    HDRSKY_CHECK_LAUNCH();
    return HDRSKY_OK;
  }
  return HDRSKY_EUNSUPPORTED;     // (more than 64 channels: the staged path - up2x_fwd, blur3, dog_mid, ...)
}

int hdrsky_l1(const float* a, const float* b, size_t n, float wl, float wg, float* loss, float* da, int accumulate,
              void* stream) {
  if (!a) return HDRSKY_EINVAL;
  if ((n & 3) == 0 && n >= 4096 && n < (1ull << 33) && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)da) & 15) == 0)) {
    const unsigned n4 = (unsigned)(n >> 2);
    unsigned grid = (n4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(l1_vec4_kernel, dim3(grid), dim3(256), 0, S_(stream), (const float4*)a, (const float4*)b, n4,
                       1.f / (float)n, wl, wg, loss, (float4*)da, accumulate);
    HDRSKY_CHECK_LAUNCH();
    return HDRSKY_OK;
  }
  hipLaunchKernelGGL(l1_kernel, dim3(grid_for(n, 2048) > 512 ? 512 : grid_for(n, 2048)), dim3(256), 0, S_(stream), a, b, n, wl, wg, loss, da, accumulate);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_mse(const float* x, float target, size_t n, float wl, float wg, float* loss, float* dx, void* stream) {
  if (!x) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(mse_kernel, dim3(grid_for(n, 2048) > 512 ? 512 : grid_for(n, 2048)), dim3(256), 0, S_(stream), x, target, n, wl, wg, loss, dx);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_kl(const float* gt, const float* cmf, int B, int N, float* loss, float* dcmf, void* stream) {
  if (!gt || !cmf) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(kl_kernel, dim3(grid_for((size_t)B * N, 1024) > 512 ? 512 : grid_for((size_t)B * N, 1024)), dim3(256), 0, S_(stream), gt, cmf, B, N, loss, dcmf);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_softmax_bwd(const float* cmf, const float* dcmf, const float* z, int M, int N, float* dz, void* stream) {
  if (!cmf || !dcmf || !z || !dz) return HDRSKY_EINVAL;
  if ((N & 3) == 0 && N <= 16384 && ((((uintptr_t)cmf | (uintptr_t)dcmf | (uintptr_t)z | (uintptr_t)dz) & 15) == 0)) {
    if (N <= 4096)
      hipLaunchKernelGGL((softmax_bwd_vec_kernel<1>), dim3(M), dim3(1024), 0, S_(stream), (const float4*)cmf, (const float4*)dcmf,
                         (const float4*)z, N / 4, (float4*)dz);
    else
      hipLaunchKernelGGL((softmax_bwd_vec_kernel<4>), dim3(M), dim3(1024), 0, S_(stream), (const float4*)cmf, (const float4*)dcmf,
                         (const float4*)z, N / 4, (float4*)dz);
    HDRSKY_CHECK_LAUNCH();
    return HDRSKY_OK;
  }
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(M), dim3(256), 0, S_(stream), cmf, dcmf, z, N, dz);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_blend_bwd(const float* y_gamma, const float* alpha, const float* dyg, const float* dyl, size_t n, float* dsky,
                     float* dsun, void* stream) {
  if (!y_gamma || !alpha || !dsky || !dsun) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), y_gamma, alpha, dyg, dyl, n, dsky, dsun);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_head_bwd(const float* y_gamma, const float* alpha, const float* dyg, const float* dyl, const float* din6, const float* y_f,
                    const float* res_f, const float* y_u, const float* res_u, size_t n, float* dc_f, float* dc_u, float* dres_u,
                    void* stream) {
  if (!y_gamma || !alpha || !y_f || !res_f || !y_u || !res_u || !dc_f || !dc_u || !dres_u || (n % 3) != 0) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), y_gamma, alpha, dyg, dyl, din6, y_f, res_f, y_u,
                     res_u, n, dc_f, dc_u, dres_u);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_decoder_tail_bwd(const float* y, const float* res, const float* dy, size_t n, float* dc, float* dres,
                            void* stream) {
  if (!y || !res || !dy || !dc) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(decoder_tail_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), y, res, dy, n, dc, dres);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

/* scratch: B*P + B floats + 1 int (count of the elements equal to the maximum, zeroed here); dcmf is accumulated into */
int hdrsky_sun_rad_bwd_slices(int P) {   // pixel slices per sample of the reduce pass (scratch: B*P + B + 4 + 3*B*slices floats)
  int S = P / 4096;
  return S < 1 ? 1 : (S > 64 ? 64 : S);
}

int hdrsky_sun_rad_bwd_reduce(const float* cmf, const void* gmax_bits, const float* gamma, const float* beta, const float* drg3,
                              int B, int P, float* scratch, float* dpre, void* stream) {
  if (!cmf || !gmax_bits || !gamma || !beta || !drg3 || !scratch || !dpre || (size_t)B * P > 0x7fffffffu) return HDRSKY_EINVAL;
  float* dx = scratch;
  float* dotx = scratch + (size_t)B * P;
  int* claimed = reinterpret_cast<int*>(dotx + B);
  hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, S_(stream), (unsigned*)claimed, (size_t)1);   // (see hdrsky_zero)
  const int S = hdrsky_sun_rad_bwd_slices(P);
  float* part = dotx + B + 4;                       // behind the record (dotx[B], tie count): 3 * B * S floats
  hipLaunchKernelGGL(sun_rad_bwd_kernel, dim3(B * S), dim3(P / S >= 2048 ? 1024 : 256), 0, S_(stream), cmf,
                     (const unsigned int*)gmax_bits, gamma, beta, drg3, P, dx, dpre, dotx, B * P, claimed, S, part);
  if (S > 1) hipLaunchKernelGGL(sun_rad_bwd_fin_kernel, dim3(B), dim3(64), 0, S_(stream), part, S, gamma, beta, dpre, dotx);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_sun_rad_bwd_apply(const float* cmf, const void* gmax_bits, const float* scratch, const float* rec, int nrec, int B,
                             int P, float* dcmf, void* stream) {
  if (!cmf || !gmax_bits || !scratch || !rec || nrec < 1 || !dcmf) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(sun_rad_bwd_cmf_kernel, dim3(grid_for((size_t)B * P)), dim3(256), 0, S_(stream), cmf,
                     (const unsigned int*)gmax_bits, scratch, rec, nrec, B, P, dcmf);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_sun_rad_bwd(const float* cmf, const void* gmax_bits, const float* gamma, const float* beta, const float* drg3,
                       int B, int P, float* scratch, float* dpre, float* dcmf, void* stream) {
  if (!dcmf) return HDRSKY_EINVAL;
  const int rc = hdrsky_sun_rad_bwd_reduce(cmf, gmax_bits, gamma, beta, drg3, B, P, scratch, dpre, stream);
  if (rc != HDRSKY_OK) return rc;
  return hdrsky_sun_rad_bwd_apply(cmf, gmax_bits, scratch, scratch + (size_t)B * P, 1, B, P, dcmf, stream);
}

int hdrsky_dense_heads_bwd(const float* x, const float* scale, const float* shift, float slope, int B, int F, int C,
                           const float* kg, const float* kb, const float* dpre, float* dact, float* dkg, float* dkb,
                           float* dbg, float* dbb, void* stream) {
  if (!x || !kg || !kb || !dpre || !dact || !dkg || !dkb || !dbg || !dbb) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(dense_heads_bwd_kernel, dim3(cdiv(F, 256)), dim3(256), 0, S_(stream), x, scale, shift, slope, B, F, C,
                     kg, kb, dpre, dact, dkg, dkb, dbg, dbb);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_slice_channels(const float* x, size_t npix, int C, int c_off, int c_take, float scale, int accumulate,
                          float* out, void* stream) {
  if (!x || !out || c_off + c_take > C) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(slice_channels_kernel, dim3(grid_for(npix * c_take)), dim3(256), 0, S_(stream), x, npix, C, c_off,
                     c_take, scale, accumulate, out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_pad_channels(const float* x, size_t npix, int C, int Cpad, float* out, void* stream) {
  if (!x || !out || C <= 0 || Cpad < C) return HDRSKY_EINVAL;
  if (npix == 0) return HDRSKY_OK;
  hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_for(npix * Cpad)), dim3(256), 0, S_(stream), x, npix, C, Cpad, out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_concat2(const float* a, int Ca, const float* b, int Cb, size_t npix, float* out, void* stream) {
  if (!a || !b || !out) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(concat2_kernel, dim3(grid_for(npix * (Ca + Cb))), dim3(256), 0, S_(stream), a, Ca, b, Cb, npix, out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_act_bf16(const float* x, int x_bf16, int B, int HW, int C, int in_mode, const float* in_scale, const float* in_shift, int ss_bstride,
                    const float* in_part, int in_nparts, const float* gamma, const float* beta, float eps, float slope,
                    void* y_bf16, void* stream) {
  if (!x || !y_bf16 || B <= 0 || HW <= 0 || C <= 0 || (C & 7) || C > 1024) return HDRSKY_EINVAL;
  if (in_mode == HDRSKY_IN_AFFINE && (!in_scale || !in_shift)) return HDRSKY_EINVAL;
  if (in_mode == HDRSKY_IN_PARTIALS && (!in_part || !gamma || !beta || in_nparts <= 0)) return HDRSKY_EINVAL;
  if ((size_t)HW * (C >> 3) > 0x7fffffffu) return HDRSKY_EINVAL;
  int bps = cdiv(HW * (C >> 3), 256 * 4);          // ~4 items per thread, all in flight at once (16 behind a dependent-load
  if (bps < 1) bps = 1;                            // prologue: 63 us for 17 MB on 128 blocks)
  hipLaunchKernelGGL(act_bf16_kernel, dim3(B * bps), dim3(256), 0, S_(stream), x, HW, C, in_mode, in_scale, in_shift, ss_bstride,
                     in_part, in_nparts, gamma, beta, eps, slope, (uint4*)y_bf16, bps, x_bf16 ? 1 : 0);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_concat_rows4(const float* s0, int w0, const float* s1, int w1, const float* s2, int w2, const float* s3, int w3,
                        int M, float* out, void* stream) {
  if (!s0 || !s1 || !s2 || !s3 || !out || M <= 0 || w0 <= 0 || w1 <= 0 || w2 <= 0 || w3 <= 0 || ((w0 | w1 | w2 | w3) & 3) ||
      (((uintptr_t)s0 | (uintptr_t)s1 | (uintptr_t)s2 | (uintptr_t)s3 | (uintptr_t)out) & 15))
    return HDRSKY_EINVAL;
  ConcatRows4 j{{(const float4*)s0, (const float4*)s1, (const float4*)s2, (const float4*)s3}, {w0 / 4, w1 / 4, w2 / 4, w3 / 4}};
  hipLaunchKernelGGL(concat_rows4_kernel, dim3(grid_for((size_t)M * (w0 + w1 + w2 + w3) / 4)), dim3(256), 0, S_(stream), j, M,
                     (float4*)out);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_vgg_pre(const float* x, size_t n, float* y, void* stream) {
  if (!x || !y || (n % 3) != 0) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(vgg_pre_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), x, n, y);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

__global__ void __launch_bounds__(256) flip3_kernel(const float* __restrict__ x, size_t npix, float* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    const float a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
    y[3 * i] = c; y[3 * i + 1] = b; y[3 * i + 2] = a;
  }
}

int hdrsky_flip_rgb(const float* x, size_t npix, float* y, void* stream) {
  if (!x || !y) return HDRSKY_EINVAL;
  if (npix == 0) return HDRSKY_OK;
  hipLaunchKernelGGL(flip3_kernel, dim3(grid_for(npix)), dim3(256), 0, S_(stream), x, npix, y);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_axpby(const float* a, float sa, const float* b, float sb, size_t n, float* y, void* stream) {
  if (!a || !y) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, S_(stream), a, sa, b, sb, n, y);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_fc_wgrad(const float* x, const float* dy, int M, int K, int N, int accumulate, float* dw, float* db,
                    void* stream) {
  if (!x || !dy || !dw || M <= 0 || M > 32 || (K & 7) || (N & 3)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(fc_wgrad_kernel, dim3(cdiv(N / 4, 256), K / 8), dim3(256), 0, S_(stream), x, dy, M, K, N, accumulate, dw, db);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_zero(void* p, size_t nbytes, void* stream) {
  if (!p) return HDRSKY_EINVAL;
  if (nbytes == 0) return HDRSKY_OK;
  // A kernel, not hipMemsetAsync: memset nodes captured into the per-segment hipGraphs of the training step did their
  // job in the FIRST replay only (ROCm 7.2) - from the second replay on, accumulators that are cleared this way (the
  // gradient reaching the res stack, the arg-max claim word of the sun-radiance backward) kept stale or foreign contents
  // and the captured step diverged from the eager one (tests/test_train_gpu.py::test_captured_replays_match_eager_steps).
  if ((reinterpret_cast<uintptr_t>(p) & 3) || (nbytes & 3)) {      // odd-sized / unaligned buffers: byte stores
    hipLaunchKernelGGL(zero_bytes_kernel, dim3(grid_for(nbytes, 1024)), dim3(256), 0, S_(stream), (unsigned char*)p, nbytes);
    HDRSKY_CHECK_LAUNCH();
    return HDRSKY_OK;
  }
  const size_t nw = nbytes / 4;
  hipLaunchKernelGGL(zero_words_kernel, dim3(grid_for(nw, 1024)), dim3(256), 0, S_(stream), (unsigned*)p, nw);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_adam(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float beta1, float beta2, float eps,
                float gscale, void* stream) {
  if (!w || !g || !m || !v || (n & 3)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4, 1024)), dim3(256), 0, S_(stream), w, g, m, v, n / 4, lr_t, beta1, beta2, eps, gscale);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_rmsprop(float* w, const float* g, float* ms, size_t n, float lr, float rho, float eps, float gscale,
                   void* stream) {
  if (!w || !g || !ms || (n & 3)) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n / 4, 1024)), dim3(256), 0, S_(stream), w, g, ms, n / 4, lr, rho, eps, gscale);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_rmsprop2(float* w1, const float* g1, float* ms1, size_t n1, float* w2, const float* g2, float* ms2, size_t n2, float lr,
                    float rho, float eps, float gscale, void* stream) {
  if (!w1 || !g1 || !ms1 || !w2 || !g2 || !ms2 || (n1 & 3) || (n2 & 3) || n1 == 0 || n2 == 0) return HDRSKY_EINVAL;
  const int nb1 = grid_for(n1 / 4, 1024), nb2 = grid_for(n2 / 4, 1024);
  hipLaunchKernelGGL(rmsprop2_kernel, dim3(nb1 + nb2), dim3(256), 0, S_(stream), w1, g1, ms1, n1 / 4, nb1, w2, g2, ms2, n2 / 4, lr, rho,
                     eps, gscale, hdrsky_hooks().opt_nt);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

int hdrsky_rmsprop_fc(float* w, const float* g, float* ms, int K, int N, float lr, float rho, float eps, float gscale,
                      void* packed_hi, void* natural_hi, void* stream) {
  if (!w || !g || !ms || !packed_hi || (K & 7) || N <= 0) return HDRSKY_EINVAL;
  hipLaunchKernelGGL(rmsprop_fc_kernel, dim3(grid_for((size_t)(K >> 3) * N, 256)), dim3(256), 0, S_(stream), w, g, ms, K, N,
                     lr, rho, eps, gscale, (uint4*)packed_hi, (unsigned short*)natural_hi);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

}  // extern "C"
