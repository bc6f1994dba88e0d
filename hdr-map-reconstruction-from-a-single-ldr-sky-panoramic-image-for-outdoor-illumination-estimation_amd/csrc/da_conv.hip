// Distortion-aware panoramic convolution (distortion_aware_ops.py:5-270; deconv2d :272-542 = bilinear resize + this).
//
// Reference data flow: per-row spherical sampling offsets -> four tf.gather_nd materialisations of [B,h,w,k*k,C]
// -> weighted sum -> [B, h*w, k*k*C] x [k*k*C, F] matmul + bias.  Here the im2col tensor never exists: for each
// filter tap a workgroup gathers its 64 output pixels' bilinear samples straight from the (L2-resident) NHWC input
// into a double-buffered bf16 LDS tile (exact reference arithmetic: float32 coordinates, clamp in y, 360-degree wrap
// in x applied once to the coordinate and once to the corner indices, weights from the UNWRAPPED corner indices),
// and feeds it to v_mfma_f32_16x16x32_bf16 against the packed filter (row = tap*C + c, the HWIO order).
#include <cmath>
#include <cstdlib>

#include <atomic>

#include "common.h"
#include "hooks.h"

namespace {

struct DaArgs {
  const float* x;
  const uint4* whi;
  const uint4* wlo;
  const float* bias;
  const float* offs;  // [h][k*k][2] (y, x), identical for every column
  float* y;
  float* stats;       // optional [B][tiles per sample][2][Cout] (sum, sum of squares) of y per 64-pixel tile, like the conv epilogue
  int B, H, W, Cin, Cout, Npad, ksize, k2, pad, in_h, in_w, cin32, nblocks, tiles_x;
  int tab_off, use_tab;     // LDS byte offset of the per-(pixel, tap) sample table behind the A tiles; 0: computed per item
  int tpr, nrounds;         // filter taps blended per barrier round (their channels side by side in the A tile), rounds
  // general sample table (data gradient: the TRANSPOSE of the gather is again a weighted gather, with up to KM source
  // pixels per (pixel, tap)): gidx / gw [H*W][k*k][KM] = source pixel index (row-major, -1 = none) and weight
  const int* gidx;
  const float* gw;
  // region variant (da_region_kernel): first source row of every group of grp_tiles 64-pixel tiles (host table) and the
  // number of source rows staged (>= what any group reads); LDS byte offsets of the table ring and of the region
  const int* row_lo;
  int src_rows, tb_off, reg_off, nq_sh, grp_tiles, groups_x, so_off, rt_cap, w_sh;
  int ncg, cgc, cg32_sh;    // channel groups of the region kernel: count, channels each, log2(cgc / 32)
  int tm;                   // pixels per tile of the region kernel (64, or 32 when 64-pixel tiles leave half the chip idle)
  int nparts;               // statistics slots per sample: one per 64-pixel tile
};

// Workgroup = NWV waves: a tile of 64 consecutive output pixels of one sample (row-major, so it spans several rows
// when W < 64) x NWV*16 filters (wave w owns filters [16w, 16w+16): all 128 filters of a res-block conv in one
// workgroup, so the gather is not repeated per filter block).  Per filter tap: every thread gathers the four corner
// pixels of its (pixel, 8-channel) items into REGISTERS one tap ahead - the loads of tap t+1 are in flight while the
// MFMAs of tap t run - then blends, rounds to bf16 and writes the tap's A tile into the other LDS buffer.
// KM = source pixels per (pixel, tap): 4 (the forward's bilinear corners) or 8 (general table: the data gradient).
template <bool PRECISE, int NWV, int IMAX, int KM>
__global__ void __launch_bounds__(NWV * 64) da_conv_kernel(const DaArgs a) {
  constexpr int TM = 64, NT = NWV * 64;
  constexpr int MAXROWS = 5;                                   // image rows a 64-pixel tile can span (W >= 16)
  // IMAX: (pixel, chunk) items per thread and tap: Cin <= 8*IMAX*NT/TM
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, lr = lane & 15;
  const int nq = a.Cin >> 3;                                  // 8-channel groups of one tap
  const int nqr = nq * a.tpr;                                 // ... of one round: tpr taps side by side (a layer with few
                                                              // input channels would otherwise pay a barrier per MFMA)
  const int plane = TM + 1;                                   // 16-byte units per channel-chunk plane
  const int buf_units = nqr * plane * (PRECISE ? 2 : 1);
  uint4* sA = reinterpret_cast<uint4*>(smem);

  int bid = blockIdx.x;
  const int nb = bid % a.nblocks; bid /= a.nblocks;
  const int tile = bid % a.tiles_x, b = bid / a.tiles_x;      // tiles_x = tiles per sample
  const int p0 = tile * TM, n0 = nb * (NWV * 16);
  const int npix = a.H * a.W;
  const int nitems = TM * nqr;

  f32x4_t acc[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) acc[mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // (a workgroup may be wider than the filter image - the data gradient of a layer with few input channels uses 8 waves
  // for their gather registers: the waves beyond the image read column block 0 and store nothing)
  const int wcol = (n0 + wave * 16 < a.Npad) ? n0 + wave * 16 : 0;
  const uint4* wlh = a.whi + (size_t)kq * a.Npad + wcol + lr;
  const uint4* wll = PRECISE ? a.wlo + (size_t)kq * a.Npad + wcol + lr : nullptr;
  const float* xb = a.x + (size_t)b * npix * a.Cin;

  // The sampling offsets of the image rows this tile covers (k*k pairs per row), staged once: the per-tap gather then
  // starts from an LDS read instead of a dependent global load in front of the corner loads.
  __shared__ float s_off[MAXROWS * 2 * 128];
  const int row0 = p0 / a.W;
  if constexpr (KM == 4) {
    const int nrow = min(p0 + TM - 1, npix - 1) / a.W - row0 + 1;
    for (int i = tid; i < nrow * a.k2 * 2; i += NT) s_off[i] = a.offs[(size_t)row0 * a.k2 * 2 + i];
    __syncthreads();
  }

  constexpr int CBMAX = IMAX * NWV / 4;                 // 32-channel k-steps per tap (Cin <= 8 * IMAX * NWV)
  constexpr bool BPF = CBMAX <= 4;                      // filter fragments prefetched a tap ahead where registers allow
  uint4 bch[BPF ? CBMAX : 1], bcl[BPF && PRECISE ? CBMAX : 1], bnh[BPF ? CBMAX : 1], bnl[BPF && PRECISE ? CBMAX : 1];
#pragma unroll
  for (int cb = 0; cb < (BPF ? CBMAX : 1); ++cb) { bch[cb] = bnh[cb] = uint4{0, 0, 0, 0}; }
#pragma unroll
  for (int cb = 0; cb < (BPF && PRECISE ? CBMAX : 1); ++cb) { bcl[cb] = bnl[cb] = uint4{0, 0, 0, 0}; }
  // Per-(pixel, tap) sample table, built once per workgroup when it fits into LDS: the four corner offsets (floats from
  // the sample's base, 0 for corners in the zero padding) and bilinear weights.  The 8-channel items of one pixel share
  // it, so the coordinate arithmetic (float32, reference order: da_tap) is done once per pixel and tap instead of once
  // per item and tap - the per-tap rounds of this kernel are instruction-issue bound.
  int4* tabO = reinterpret_cast<int4*>(smem + a.tab_off);
  float4* tabW = reinterpret_cast<float4*>(smem + a.tab_off + TM * a.k2 * 16);
  auto sample = [&](int m, int tn, int (&o)[4], float (&w)[4]) {
    const int pix = p0 + m;
    const bool live = pix < npix;
    const int oy = live ? pix / a.W : row0, ox = live ? pix % a.W : 0;
    const float off_y = s_off[((oy - row0) * a.k2 + tn) * 2], off_x = s_off[((oy - row0) * a.k2 + tn) * 2 + 1];
    // base grid = VALID patches of the padded meshgrid (:152-168): padded coordinate of tap (ty,tx) at (oy,ox)
    const Tap4 s = da_tap((float)(oy + tn / a.ksize), (float)(ox + tn % a.ksize), off_y, off_x, a.in_h, a.in_w);
    const int ys[4] = {s.y0, s.y0, s.y1, s.y1}, xs[4] = {s.x0, s.x1, s.x0, s.x1};
    const float ws[4] = {s.w0, s.w1, s.w2, s.w3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = ys[k] - a.pad, xx = xs[k] - a.pad;      // back to un-padded coordinates; border = zeros
      const bool in = live && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
      o[k] = in ? (yy * a.W + xx) * a.Cin : 0;
      w[k] = in ? ws[k] : 0.f;
    }
  };
  if (KM == 4 && a.use_tab) {
    for (int e = tid; e < TM * a.k2; e += NT) {
      int o[4]; float w[4];
      sample(e % TM, e / TM, o, w);
      tabO[e] = int4{o[0], o[1], o[2], o[3]};
      tabW[e] = float4{w[0], w[1], w[2], w[3]};
    }
    __syncthreads();
  }

  float cr0[IMAX][KM][8], cw0[IMAX][KM];    // source pixels of this thread's items for the tap in flight / their weights

  auto gather = [&](int rnd, float (&cr)[IMAX][KM][8], float (&cw)[IMAX][KM]) {
#pragma unroll
    for (int it = 0; it < IMAX; ++it) {
      const int i = it * NT + tid;
      if (i < nitems) {
        const int m = i / nqr, qr = i % nqr;
        const int tsub = qr / nq, q = qr - tsub * nq;
        const bool tap_ok = rnd * a.tpr + tsub < a.k2;            // the last round of a 7x7 layer is partly empty
        const int tn = tap_ok ? rnd * a.tpr + tsub : a.k2 - 1;
        int o[KM]; float w[KM];
        if constexpr (KM != 4) {
          const int pix = min(p0 + m, npix - 1);
          const bool live = p0 + m < npix;
          const int* gi = a.gidx + ((size_t)pix * a.k2 + tn) * KM;
          const float* gwp = a.gw + ((size_t)pix * a.k2 + tn) * KM;
#pragma unroll
          for (int k = 0; k < KM; ++k) {
            const int si = gi[k];
            o[k] = (live && si >= 0) ? si * a.Cin : 0;
            w[k] = (live && si >= 0) ? gwp[k] : 0.f;
          }
        } else if (a.use_tab) {
          const int4 to = tabO[tn * TM + m];
          const float4 tw = tabW[tn * TM + m];
          o[0] = to.x; o[1] = to.y; o[2] = to.z; o[3] = to.w;
          w[0] = tw.x; w[1] = tw.y; w[2] = tw.z; w[3] = tw.w;
        } else {
          int o4[4]; float w4[4];
          sample(m, tn, o4, w4);
#pragma unroll
          for (int k = 0; k < 4; ++k) { o[k] = o4[k]; w[k] = w4[k]; }
        }
#pragma unroll
        for (int k = 0; k < KM; ++k) {
          const float* pp = xb + o[k] + q * 8;
          const float4 lo = *reinterpret_cast<const float4*>(pp), hi = *reinterpret_cast<const float4*>(pp + 4);
          cr[it][k][0] = lo.x; cr[it][k][1] = lo.y; cr[it][k][2] = lo.z; cr[it][k][3] = lo.w;
          cr[it][k][4] = hi.x; cr[it][k][5] = hi.y; cr[it][k][6] = hi.z; cr[it][k][7] = hi.w;
          cw[it][k] = tap_ok ? w[k] : 0.f;
        }
      }
    }
  };
  const int ksteps = a.k2 * a.cin32;      // 32-channel k-steps of the whole filter, tap-major
  const int spr = a.tpr * a.cin32;        // ... per round
  auto load_b = [&](int rnd) {   // this wave's filter fragments of round rnd (registers, one round ahead)
#pragma unroll
    for (int cb = 0; cb < CBMAX; ++cb)
      if (cb < spr && rnd * spr + cb < ksteps) {
        const size_t o = (size_t)((rnd * spr + cb) * 4) * a.Npad;
        bnh[cb] = wlh[o];
        if (PRECISE) bnl[cb] = wll[o];
      }
  };
  // one round: blend + store tap t from the register set, refill it with tap t+1 (in flight during the MFMAs of tap t)
  auto round = [&](int t, float (&cr)[IMAX][KM][8], float (&cw)[IMAX][KM]) {
    uint4* buf = sA + (t & 1) * buf_units;
#pragma unroll
    for (int it = 0; it < IMAX; ++it) {
      const int i = it * NT + tid;
      if (i < nitems) {
        const int m = i / nqr, q = i % nqr;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = cw[it][0] * cr[it][0][j] + cw[it][1] * cr[it][1][j] + cw[it][2] * cr[it][2][j] + cw[it][3] * cr[it][3][j];
          if constexpr (KM == 8)
            v[j] += cw[it][4] * cr[it][4][j] + cw[it][5] * cr[it][5][j] + cw[it][6] * cr[it][6][j] + cw[it][7] * cr[it][7][j];
        }
        uint4 h8, l8;
        pack8<PRECISE>(v, h8, l8);
        buf[q * plane + m] = h8;
        if (PRECISE) buf[nqr * plane + q * plane + m] = l8;
      }
    }
    __syncthreads();   // tile t staged; also: every wave is past the MFMAs of tile t-1, so buffer (t+1)&1 is free
    if (t + 1 < a.nrounds) gather(t + 1, cr, cw);
    if (BPF && t + 1 < a.nrounds) load_b(t + 1);
    // ---- MFMA: Cin/32 k-steps, 4 pixel fragments x this wave's 16 filters -----------------------------------
#pragma unroll
    for (int cb = 0; cb < CBMAX; ++cb) {
      if (cb >= spr || t * spr + cb >= ksteps) break;
      uint4 bh, bl = uint4{0, 0, 0, 0};
      if (BPF) { bh = bch[cb]; if (PRECISE) bl = bcl[cb]; }
      else {
        const size_t o = (size_t)((t * spr + cb) * 4) * a.Npad;
        bh = wlh[o];
        if (PRECISE) bl = wll[o];
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const uint4 ah = buf[(cb * 4 + kq) * plane + mi * 16 + lr];
        if (PRECISE) {
          const uint4 al = buf[nqr * plane + (cb * 4 + kq) * plane + mi * 16 + lr];
          acc[mi] = mfma16(al, bh, acc[mi]);
          acc[mi] = mfma16(ah, bl, acc[mi]);
        }
        acc[mi] = mfma16(ah, bh, acc[mi]);
      }
    }
    if (BPF) {
#pragma unroll
      for (int cb = 0; cb < CBMAX; ++cb) { bch[cb] = bnh[cb]; if (PRECISE) bcl[cb] = bnl[cb]; }
    }
  };

  gather(0, cr0, cw0);
  if (BPF) {
    load_b(0);
#pragma unroll
    for (int cb = 0; cb < CBMAX; ++cb) { bch[cb] = bnh[cb]; if (PRECISE) bcl[cb] = bnl[cb]; }
  }
  for (int t = 0; t < a.nrounds; ++t) round(t, cr0, cw0);
  // ---- epilogue: + bias, store -------------------------------------------------------------------------------
  const int n = n0 + wave * 16 + lr;
  float s1 = 0.f, s2 = 0.f;
  if (n < a.Cout) {
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pix = p0 + mi * 16 + kq * 4 + j;
        if (pix < npix) {
          const float v = acc[mi][j] + bv;
          a.y[((size_t)b * npix + pix) * a.Cout + n] = v;
          s1 += v; s2 += v * v;
        }
      }
  }
  if (a.stats) {   // InstanceNorm partials of this tile: lanes lr, lr+16, lr+32, lr+48 hold the same channel
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (kq == 0 && n < a.Cout) {
      float* dst = a.stats + ((size_t)(b * a.nparts + tile) * 2) * a.Cout + n;
      dst[0] = s1; dst[a.Cout] = s2;
    }
  }
}

// Region variant of the kernel above (BF16 compute mode).  The offsets of this layer depend on the image ROW only, so the
// samples of a 64-pixel tile fall into a few consecutive source rows (5-6 for the 3x3 layers, 13-14 for 7x7 at 32x128:
// kernels.da_row_lo computes the span from the host sample table).  The workgroup stages those rows ONCE (fp32 -> bf16,
// a linear copy) and then takes every corner of every tap from LDS instead of from L2: the legacy kernel pulls
// k*k x 4 corner rows per pixel and channel through the vector memory path (7x7, 32 channels, B = 32: 3.3 GB per launch).
// Per barrier round: (a) gather + blend + store the round's A tile from the region via a small per-(pixel, tap) table
// (region pixel index as u16 + fp32 weight per source) that was written one round earlier, (b) write the NEXT round's
// table (KM = 4: da_tap, exact reference arithmetic; KM = 8: the transposed table from global memory, fetched a round
// ahead), barrier, (c) the MFMAs of the round.  Sources are rounded to bf16 before the blend (the legacy kernel blends
// fp32 sources); the blend itself is fp32 and its result is rounded once more, as there.
template <int NWV, int KM, int CBMAX, int TM>
__global__ void __launch_bounds__(NWV * 64) da_region_kernel(const DaArgs a) {
  constexpr int NT = NWV * 64, TSH = TM == 64 ? 6 : 5, MF = TM / 16;   // TM = 64 or 32 pixels per tile
  constexpr int EMAX = 2;                                       // CBMAX: 32-channel k-steps per round (register sets)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, lr = lane & 15;
  // channel groups: when the rows of a tile do not fit LDS with all channels, the region holds cgc of them at a time and
  // the rounds run once per group (the accumulators carry over; k-step = tap * Cin/32 + the group's 32-channel block)
  const int nq = a.cgc >> 3, nqr = nq * a.tpr, plane = TM + 1, buf_units = nqr * plane;
  uint4* sA = reinterpret_cast<uint4*>(smem);
  const int nent = TM * a.tpr;                                  // table entries per round
  unsigned char* sTb = smem + a.tb_off;                         // [2] x { u16 idx[nent][KM], float w[nent][KM] }
  const int tb_bytes = nent * KM * 6;
  const uint4* sReg = reinterpret_cast<const uint4*>(smem + a.reg_off);

  int bid = blockIdx.x;
  const int nb = bid % a.nblocks; bid /= a.nblocks;
  const int grp = bid % a.groups_x, b = bid / a.groups_x;     // group = grp_tiles consecutive tiles sharing one region
  const int n0 = nb * (NWV * 16);
  const int npix = a.H * a.W;
  const int ylo = a.row_lo[TM == 64 ? grp : grp >> 1];          // a 32-pixel tile stages the rows of the 64-pixel tile it lies in
  const int nrows = min(a.src_rows, a.H - ylo);
  const int regpix = nrows * a.W;
  const int tile_end = min((grp + 1) * a.grp_tiles, a.tiles_x);

  f32x4_t acc[MF];
  const bool wave_live = n0 + wave * 16 < a.Npad;     // waves beyond the filter image only help with the gather
  const int wcol = wave_live ? n0 + wave * 16 : 0;
  const uint4* wlh = a.whi + (size_t)kq * a.Npad + wcol + lr;

  // row halves of the sample positions of the tile's rows (da_tap_y: region row * W or -1, the two row weights), the
  // column offsets and tap % k - the per-entry work left is da_tap_x, four products and four index sums
  int4* sRT = reinterpret_cast<int4*>(smem + a.so_off);
  float* sOx = reinterpret_cast<float*>(smem + a.so_off + a.rt_cap * 16);
  int* sKx = reinterpret_cast<int*>(smem + a.so_off + a.rt_cap * 20);
  // ---- the source rows of this group (channels [cg * cgc, (cg + 1) * cgc)): fp32 -> bf16 ---------------------------
  auto stage_region = [&](int cg) {
    const float* src = a.x + ((size_t)b * npix + (size_t)ylo * a.W) * a.Cin + cg * a.cgc;
    uint4* dst = reinterpret_cast<uint4*>(smem + a.reg_off);
    const int nunits = regpix * nq;
    auto addr = [&](int u) { return src + (size_t)(u >> a.nq_sh) * a.Cin + (u & (nq - 1)) * 8; };
    int u = tid;
    for (; u + 3 * NT < nunits; u += 4 * NT) {
      float4 lo[4], hi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* pp = addr(u + k * NT);
        lo[k] = *reinterpret_cast<const float4*>(pp); hi[k] = *reinterpret_cast<const float4*>(pp + 4);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v[8] = {lo[k].x, lo[k].y, lo[k].z, lo[k].w, hi[k].x, hi[k].y, hi[k].z, hi[k].w};
        uint4 h8, l8;
        pack8<false>(v, h8, l8);
        dst[u + k * NT] = h8;
      }
    }
    for (; u < nunits; u += NT) {
      const float* pp = addr(u);
      const float4 lo = *reinterpret_cast<const float4*>(pp), hi = *reinterpret_cast<const float4*>(pp + 4);
      const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      uint4 h8, l8;
      pack8<false>(v, h8, l8);
      dst[u] = h8;
    }
  };
  if (a.ncg == 1) stage_region(0);
  uint4 bch[CBMAX], bnh[CBMAX];
  const int cg32 = a.cgc >> 5, spr = a.tpr * cg32;               // 32-channel k-steps of a group / of a round
  const int nitems = TM * nqr;
  for (int tile = grp * a.grp_tiles; tile < tile_end; ++tile) {
  const int p0 = tile * TM;
  const int row0 = p0 / a.W;
#pragma unroll
  for (int mi = 0; mi < MF; ++mi) acc[mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if constexpr (KM == 4) {
    const int nrow = min(p0 + TM - 1, npix - 1) / a.W - row0 + 1;
    for (int i = tid; i < nrow * a.k2; i += NT) {
      const int r = i / a.k2, tn = i - r * a.k2;
      const float2 o = *reinterpret_cast<const float2*>(a.offs + ((size_t)row0 * a.k2 + i) * 2);
      const TapAxis ty = da_tap_y((float)(row0 + r + tn / a.ksize), o.x, a.in_h);
      const int r0 = ty.i0 - a.pad - ylo, r1 = ty.i1 - a.pad - ylo;
      sRT[i] = int4{(r0 >= 0 && r0 < nrows) ? r0 * a.W : -1, (r1 >= 0 && r1 < nrows) ? r1 * a.W : -1,
                    __builtin_bit_cast(int, ty.wa), __builtin_bit_cast(int, ty.wb)};
      sOx[i] = o.y;
    }
    for (int tn = tid; tn < a.k2; tn += NT) sKx[tn] = tn % a.ksize;
    __syncthreads();     // visible to the table writers
  }

  for (int cg = 0; cg < a.ncg; ++cg) {
  if (a.ncg > 1) stage_region(cg);               // (the barrier behind table 0 below publishes it)
  // ---- the per-round sample table ---------------------------------------------------------------------------------
  int pend_i[KM == 8 ? EMAX : 1][8];
  float pend_w[KM == 8 ? EMAX : 1][8];
  auto table_fetch = [&](int rnd) {            // KM = 8: global loads of round rnd's entries into registers
    if constexpr (KM == 8) {
#pragma unroll
      for (int k = 0; k < EMAX; ++k) {
        const int e = k * NT + tid;
        if (e < nent) {
          const int m = e & (TM - 1), tn = min(rnd * a.tpr + (e >> TSH), a.k2 - 1);
          const int pix = min(p0 + m, npix - 1);
          const int4* gi = reinterpret_cast<const int4*>(a.gidx + ((size_t)pix * a.k2 + tn) * 8);
          const float4* gwp = reinterpret_cast<const float4*>(a.gw + ((size_t)pix * a.k2 + tn) * 8);
          const int4 i0 = gi[0], i1 = gi[1];
          const float4 w0 = gwp[0], w1 = gwp[1];
          pend_i[k][0] = i0.x; pend_i[k][1] = i0.y; pend_i[k][2] = i0.z; pend_i[k][3] = i0.w;
          pend_i[k][4] = i1.x; pend_i[k][5] = i1.y; pend_i[k][6] = i1.z; pend_i[k][7] = i1.w;
          pend_w[k][0] = w0.x; pend_w[k][1] = w0.y; pend_w[k][2] = w0.z; pend_w[k][3] = w0.w;
          pend_w[k][4] = w1.x; pend_w[k][5] = w1.y; pend_w[k][6] = w1.z; pend_w[k][7] = w1.w;
        }
      }
    }
  };
  auto table_write = [&](int rnd) {            // entries of round rnd -> ring buffer rnd & 1
    unsigned short* ti = reinterpret_cast<unsigned short*>(sTb + (rnd & 1) * tb_bytes);
    float* tw = reinterpret_cast<float*>(sTb + (rnd & 1) * tb_bytes + nent * KM * 2);
#pragma unroll
    for (int k = 0; k < EMAX; ++k) {
      const int e = k * NT + tid;
      if (e < nent) {
        const int m = e & (TM - 1), tsub = e >> TSH;
        const bool tap_ok = rnd * a.tpr + tsub < a.k2;
        const int tn = tap_ok ? rnd * a.tpr + tsub : a.k2 - 1;
        const bool live = p0 + m < npix;
        unsigned short oi[KM];
        float ow[KM];
        if constexpr (KM == 4) {
          const int pix = live ? p0 + m : p0;
          const int ri = ((pix >> a.w_sh) - row0) * a.k2 + tn, ox = pix & (a.W - 1);      // W is a power of two
          const int4 rt = sRT[ri];
          const TapAxis tx = da_tap_x((float)(ox + sKx[tn]), sOx[ri], a.in_w);
          const float wya = __builtin_bit_cast(float, rt.z), wyb = __builtin_bit_cast(float, rt.w);
          const int rys[4] = {rt.x, rt.x, rt.y, rt.y}, xs[4] = {tx.i0 - a.pad, tx.i1 - a.pad, tx.i0 - a.pad, tx.i1 - a.pad};
          const float ws[4] = {wya * tx.wa, wya * tx.wb, wyb * tx.wa, wyb * tx.wb};        // = da_tap's w0..w3
#pragma unroll
          for (int c = 0; c < 4; ++c) {     // outside the image = the layer's zero padding
            const bool in = live && tap_ok && rys[c] >= 0 && xs[c] >= 0 && xs[c] < a.W;
            oi[c] = (unsigned short)(in ? rys[c] + xs[c] : 0);
            ow[c] = in ? ws[c] : 0.f;
          }
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int si = pend_i[k][c] - ylo * a.W;
            const bool in = live && tap_ok && pend_i[k][c] >= 0 && si >= 0 && si < regpix;
            oi[c] = (unsigned short)(in ? si : 0);
            ow[c] = in ? pend_w[k][c] : 0.f;
          }
        }
#pragma unroll
        for (int c = 0; c < KM; c += 4) {
          *reinterpret_cast<uint2*>(ti + (size_t)e * KM + c) =
              uint2{(unsigned)oi[c] | ((unsigned)oi[c + 1] << 16), (unsigned)oi[c + 2] | ((unsigned)oi[c + 3] << 16)};
          *reinterpret_cast<float4*>(tw + (size_t)e * KM + c) = float4{ow[c], ow[c + 1], ow[c + 2], ow[c + 3]};
        }
      }
    }
  };

#pragma unroll
  for (int cb = 0; cb < CBMAX; ++cb) bch[cb] = bnh[cb] = uint4{0, 0, 0, 0};
  // this wave's filter fragments of the next round (fetching them two rounds ahead into a ring of three register sets was
  // tried: the registers it costs halve the taps per round, and the rounds, not the filter fetch, are what a tile pays for)
  auto load_b = [&](int rnd) {
#pragma unroll
    for (int cb = 0; cb < CBMAX; ++cb)
      if (cb < spr && rnd * a.tpr + (cb >> a.cg32_sh) < a.k2)
        bnh[cb] = wlh[(size_t)(((rnd * a.tpr + (cb >> a.cg32_sh)) * a.cin32 + cg * cg32 + (cb & (cg32 - 1))) * 4) * a.Npad];
  };

  table_fetch(0);
  table_write(0);
  if (a.nrounds > 1) table_fetch(1);
  load_b(0);
#pragma unroll
  for (int cb = 0; cb < CBMAX; ++cb) bch[cb] = bnh[cb];
  __syncthreads();                              // region + table 0 staged

  for (int t = 0; t < a.nrounds; ++t) {
    uint4* buf = sA + (t & 1) * buf_units;
    const unsigned short* ti = reinterpret_cast<const unsigned short*>(sTb + (t & 1) * tb_bytes);
    const float* tw = reinterpret_cast<const float*>(sTb + (t & 1) * tb_bytes + nent * KM * 2);
#pragma unroll(KM == 4 ? 2 : 1)
    for (int i = tid; i < nitems; i += NT) {
      // item order (tap of the round, pixel, 8-channel group): nq and TM are powers of two, so no division - the
      // nq lanes of a (pixel, tap) share one table entry (LDS broadcast)
      const int e = i >> a.nq_sh, q = i & (nq - 1);
      const int m = e & (TM - 1), qr = (e >> TSH) * nq + q;
      unsigned short oi[KM];
      float ow[KM];
#pragma unroll
      for (int c = 0; c < KM; c += 4) {
        const uint2 pi = *reinterpret_cast<const uint2*>(ti + (size_t)e * KM + c);
        const float4 pw = *reinterpret_cast<const float4*>(tw + (size_t)e * KM + c);
        oi[c] = (unsigned short)(pi.x & 0xffffu); oi[c + 1] = (unsigned short)(pi.x >> 16);
        oi[c + 2] = (unsigned short)(pi.y & 0xffffu); oi[c + 3] = (unsigned short)(pi.y >> 16);
        ow[c] = pw.x; ow[c + 1] = pw.y; ow[c + 2] = pw.z; ow[c + 3] = pw.w;
      }
      uint4 src[KM];
#pragma unroll
      for (int c = 0; c < KM; ++c) src[c] = sReg[(int)oi[c] * nq + q];
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
      for (int c = 0; c < KM; ++c) {
        const unsigned w4[4] = {src[c].x, src[c].y, src[c].z, src[c].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[2 * j] += ow[c] * __builtin_bit_cast(float, w4[j] << 16);
          v[2 * j + 1] += ow[c] * __builtin_bit_cast(float, w4[j] & 0xffff0000u);
        }
      }
      uint4 h8, l8;
      pack8<false>(v, h8, l8);
      buf[qr * plane + m] = h8;
    }
    if (t + 1 < a.nrounds) table_write(t + 1);
    __syncthreads();   // A tile t and table t+1 staged; every wave is past the MFMAs of round t-1
    if (t + 2 < a.nrounds) table_fetch(t + 2);
    if (wave_live) {
      if (t + 1 < a.nrounds) load_b(t + 1);
#pragma unroll
      for (int cb = 0; cb < CBMAX; ++cb) {
        if (cb >= spr || t * a.tpr + (cb >> a.cg32_sh) >= a.k2) break;
        const uint4 bh = bch[cb];
#pragma unroll
        for (int mi = 0; mi < MF; ++mi) acc[mi] = mfma16(buf[(cb * 4 + kq) * plane + mi * 16 + lr], bh, acc[mi]);
      }
#pragma unroll
      for (int cb = 0; cb < CBMAX; ++cb) bch[cb] = bnh[cb];
    }
  }
  if (a.ncg > 1) __syncthreads();    // the next channel group overwrites the region, the table ring and the A buffers
  }   // channel groups
  // ---- epilogue: + bias, store (as da_conv_kernel) --------------------------------------------------------------
  const int n = n0 + wave * 16 + lr;
  float s1 = 0.f, s2 = 0.f;
  if (n < a.Cout) {
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pix = p0 + mi * 16 + kq * 4 + j;
        if (pix < npix) {
          const float v = acc[mi][j] + bv;
          a.y[((size_t)b * npix + pix) * a.Cout + n] = v;
          s1 += v; s2 += v * v;
        }
      }
  }
  if (a.stats) {
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (kq == 0 && n < a.Cout) {
      float* dst = a.stats + ((size_t)(b * a.nparts + tile) * 2) * a.Cout + n;   // (host: statistics only with TM = 64)
      dst[0] = s1; dst[a.Cout] = s2;
    }
  }
  __syncthreads();      // the next tile reuses s_off, the table ring and the A buffers
  }   // tiles of the group
}

// Plans the region variant for a layer whose gather source has C channels on an H x W map: taps per round, LDS layout.
// Returns the dynamic LDS bytes, or 0 when the layer does not fit (the caller then uses da_conv_kernel).
static int da_region_plan(DaArgs& a, int C, int nwv, int km, const int* spans, int B) {
  // spans (host): rows a group of 1, 2, 4, 8, 16 consecutive tiles reads (kernels.da_row_lo); a.row_lo (device): the
  // first row of every group, [5][tiles_x].  Larger groups re-read fewer source rows (a 7x7 tile of 64 pixels stages 13
  // rows of 128 pixels: 26 x its own size; 8 tiles sharing 16 rows: 4 x) - taken while the launch keeps >= 256 workgroups.
  if (!spans || !a.row_lo || (C % 32) != 0) return 0;
  const HdrskyHooks& hk = hdrsky_hooks();
  if (hk.da_region == 0) return 0;                          // switch: never the region kernels
  const int hook_level = hk.da_group;                       // tuning hook: tiles per region (-1: one)
  const int nq = C / 8, nt = nwv * 64;
  if ((nq & (nq - 1)) || (a.W & (a.W - 1))) return 0;           // the item / pixel indexing wants powers of two
  a.w_sh = __builtin_ctz(a.W);
  // 32-pixel tiles when 64-pixel tiles would leave half the CUs without a workgroup (the 8x32 / 4x16 maps at batch 32)
  a.tm = (B * a.tiles_x * a.nblocks <= 128 && hook_level <= 0) ? 32 : 64;
  if (hk.da_tm == 32 || hk.da_tm == 64) a.tm = hk.da_tm;
  if (a.stats) a.tm = 64;       // the InstanceNorm partials are per 64-pixel tile (one writer per slot)
  if (a.tm == 32) a.tiles_x = cdiv(a.H * a.W, 32);
  a.rt_cap = km == 4 ? (64 / a.W + 2) * a.k2 : 0;              // (row, tap) entries of a tile: 16 + 4 bytes each, + k2 ints
  const int so = km == 4 ? roundup(a.rt_cap * 20 + a.k2 * 4, 16) : 0;
  const int* row_lo_base = a.row_lo;
  for (int level = 4; level >= 0; --level) {
    if (a.tm == 32 && level != 0) continue;
    const int G = 1 << level, groups = cdiv(a.tiles_x, G), src_rows = spans[level];
    if (level != (hook_level >= 0 ? hook_level : 0)) continue;   // measured: groups of tiles do not pay (see DESIGN), hook only
    if (src_rows <= 0 || (size_t)src_rows * a.W > 65535) continue;   // u16 region pixel indices
    // all channels in the region when they fit; else, for the data gradient, 2 or 4 channel groups (one tile per workgroup
    // then).  Measured at [8,32,128,128]: data gradient 97 -> 75 us; the forward pass gains nothing from two groups (64 us
    // either way: twice the tables and barriers), so it keeps da_conv_kernel there.
    for (int ncg = 1; ncg <= (level == 0 && km == 8 ? 4 : 1); ncg *= 2) {
      const int cgc = C / ncg, nqg = cgc / 8;
      if ((C % ncg) != 0 || (cgc % 32) != 0) continue;
      const int region = src_rows * a.W * cgc * 2;
      for (int tpr = a.k2 < 8 ? a.k2 : 8; tpr >= 1; --tpr) {
        if (tpr * (cgc / 32) > 8 || 64 * tpr > 2 * nt) continue;   // filter prefetch registers, table entries per thread
        const int abytes = 2 * tpr * nqg * (a.tm + 1) * 16, tb = 2 * a.tm * tpr * km * 6;
        if (abytes + tb + so + region > 160 * 1024) continue;
        a.tpr = tpr; a.nrounds = cdiv(a.k2, tpr);
        a.ncg = ncg; a.cgc = cgc; a.cg32_sh = __builtin_ctz(cgc / 32); a.nq_sh = __builtin_ctz(nqg);
        a.tb_off = abytes; a.so_off = abytes + tb; a.reg_off = abytes + tb + so; a.src_rows = src_rows;
        a.grp_tiles = G; a.groups_x = groups; a.row_lo = row_lo_base + (size_t)level * a.tiles_x;
        return abytes + tb + so + region;
      }
    }
  }
  a.tiles_x = cdiv(a.H * a.W, 64);      // (the caller falls back to da_conv_kernel's 64-pixel tiles)
  return 0;
}

// HDRSKY_DA_REGION=2: fail instead of falling back to the global-memory gather (tests: proves which kernel ran)
static bool da_region_forced() { return hdrsky_hooks().da_region == 2; }

template <int NWV, int KM, int CBMAX, int TM>
static int da_region_launch_(const DaArgs& a, int grid, int lds, void* stream) {
  auto k = da_region_kernel<NWV, KM, CBMAX, TM>;
  static std::atomic<bool> set{false};
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return HDRSKY_ELAUNCH;
    set = true;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NWV * 64), lds, (hipStream_t)stream, a);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

template <int NWV, int KM>
static int da_region_launch(const DaArgs& a, int grid, int lds, void* stream) {
  if (a.tm == 32)
    return a.tpr * (a.cgc / 32) <= 4 ? da_region_launch_<NWV, KM, 4, 32>(a, grid, lds, stream) : da_region_launch_<NWV, KM, 8, 32>(a, grid, lds, stream);
  return a.tpr * (a.cgc / 32) <= 4 ? da_region_launch_<NWV, KM, 4, 64>(a, grid, lds, stream) : da_region_launch_<NWV, KM, 8, 64>(a, grid, lds, stream);
}

// ---- backward building blocks ----------------------------------------------------------------------------------
// The layer is y = G(x) W + b with G the (linear) bilinear gather [B,h,w,k*k*C].  Its gradients are
//   dW = G(x)^T dY   (a 1x1-conv weight gradient on the gathered tensor),   db = sum dY,
//   dX = G^T (dY W^T) (a 1x1 conv producing dG, then the transpose of the gather = a bilinear SCATTER with the same
//                      corner indices and weights; fp32 atomics because several samples share a corner).
struct DaGsArgs {
  const float* src;   // gather: x [B,H,W,C]            scatter: dG [B,H,W,k2*C]
  float* dst;         // gather: G [B,H,W,k2*C]         scatter: dx [B,H,W,C] (zeroed by the caller)
  const float* offs;
  int B, H, W, C, ksize, k2, pad, in_h, in_w;
};

template <bool SCATTER>
__global__ void __launch_bounds__(256) da_gather_scatter_kernel(const DaGsArgs a) {
  const int nq = a.C >> 3;
  const size_t total = (size_t)a.B * a.H * a.W * a.k2 * nq;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % nq);
    size_t r = i / nq;
    const int t = (int)(r % a.k2); r /= a.k2;
    const int ox = (int)(r % a.W); r /= a.W;
    const int oy = (int)(r % a.H);
    const int b = (int)(r / a.H);
    const float off_y = a.offs[(oy * a.k2 + t) * 2], off_x = a.offs[(oy * a.k2 + t) * 2 + 1];
    const Tap4 s = da_tap((float)(oy + t / a.ksize), (float)(ox + t % a.ksize), off_y, off_x, a.in_h, a.in_w);
    const int ys[4] = {s.y0, s.y0, s.y1, s.y1}, xs[4] = {s.x0, s.x1, s.x0, s.x1};
    const float ws[4] = {s.w0, s.w1, s.w2, s.w3};
    const size_t gidx = ((((size_t)b * a.H + oy) * a.W + ox) * a.k2 + t) * a.C + q * 8;
    float v[8];
    if (SCATTER) {
      const float4 lo = *reinterpret_cast<const float4*>(a.src + gidx), hi = *reinterpret_cast<const float4*>(a.src + gidx + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = ys[k] - a.pad, xx = xs[k] - a.pad;      // un-padded coordinates; the zero border carries no gradient
      if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
        const size_t xi = (((size_t)b * a.H + yy) * a.W + xx) * a.C + q * 8;
        if (SCATTER) {
#pragma unroll
          for (int j = 0; j < 8; ++j) atomicAdd(a.dst + xi + j, ws[k] * v[j]);
        } else {
          const float4 lo = *reinterpret_cast<const float4*>(a.src + xi), hi = *reinterpret_cast<const float4*>(a.src + xi + 4);
          v[0] += ws[k] * lo.x; v[1] += ws[k] * lo.y; v[2] += ws[k] * lo.z; v[3] += ws[k] * lo.w;
          v[4] += ws[k] * hi.x; v[5] += ws[k] * hi.y; v[6] += ws[k] * hi.z; v[7] += ws[k] * hi.w;
        }
      }
    }
    if (!SCATTER) {
      *reinterpret_cast<float4*>(a.dst + gidx) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(a.dst + gidx + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// ---- the gathered operand as a bf16 tensor ----------------------------------------------------------------------------
// G[b][p][t*C + c] = sum_m w(p,t,m) * src[b][idx(p,t,m)][c], rounded to bf16: exactly what the fused kernels above feed the
// matrix cores, written once.  With it the layer IS the reference's own formulation (distortion_aware_ops.py:107-121: gather,
// then matmul with the [k*k*C, F] kernel): a 1x1 convolution over k*k*C channels for the generic conv / weight-gradient
// kernels, whose packed filter image is the k x k filter's own (same k-step order).  Two position sources:
//   offs != NULL: the forward's four corners (da_tap, the reference's float32 arithmetic);
//   gidx / gw [H*W][k*k][km]: a sample table - the TRANSPOSED table of hdrsky_da_conv2d_dgrad (km = 8) makes this the
//   operand of the data gradient, in the tap order of the transpose_flip filter image.
// src: fp32 or bf16.  Thread = (sample, pixel, tap, 8 channels): 16-byte loads per corner, one 16-byte store; consecutive
// threads write consecutive bytes.  Measured against the fused region kernels on the 128x512 step: profiles/r04_da_mat_ab.txt.
struct DaG16Args {
  const float* src; int src_bf16;
  unsigned short* dst;
  const float* offs;
  const int* gidx; const float* gw; int km;
  int B, H, W, C, ksize, k2, pad, in_h, in_w;
  int nq, lognq, pixb;   // 8-channel groups per pixel (lognq >= 0: a power of two); pixels per workgroup
};

// Workgroup = (sample, DA_G16_PIXB consecutive pixels): the sample positions of its (pixel, tap) pairs are worked out ONCE
// (one thread each: da_tap, or a table row) into LDS; the items (pixel, tap, 8 channels) - contiguous 16-byte pieces of G in
// exactly this order - are then shared out over the threads: per item one LDS entry, up to km 16/32-byte source loads, one
// 16-byte store.  No per-item index division (the first version spent most of its time in 64-bit divisions: 105 us for the
// 75 MB of a 128-channel layer at 32x128, batch 8).
constexpr int DA_G16_KM = 8;
__device__ __forceinline__ void g16_unpack8(const uint4& u, float (&v)[8]) {   // 8 bf16 -> 8 floats
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __builtin_bit_cast(float, w[k] << 16);
    v[2 * k + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
  }
}
__global__ void __launch_bounds__(256) da_gather_bf16_kernel(const DaG16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g16_smem[];
  int* sIdx = reinterpret_cast<int*>(g16_smem);                     // [pixb * k2][km]
  const int km = a.offs != nullptr ? 4 : a.km;
  const int npt = a.pixb * a.k2;
  float* sW = reinterpret_cast<float*>(g16_smem) + npt * km;
  const int HW = a.H * a.W;
  const int pix0 = blockIdx.x * a.pixb, b = blockIdx.y;
  const int npix = min(a.pixb, HW - pix0);
  for (int e = threadIdx.x; e < npix * a.k2; e += 256) {
    const int pl = e / a.k2, t = e - pl * a.k2, pix = pix0 + pl;
    if (a.offs != nullptr) {
      const int oy = pix / a.W, ox = pix - oy * a.W;
      const float off_y = a.offs[(oy * a.k2 + t) * 2], off_x = a.offs[(oy * a.k2 + t) * 2 + 1];
      const Tap4 s = da_tap((float)(oy + t / a.ksize), (float)(ox + t % a.ksize), off_y, off_x, a.in_h, a.in_w);
      const int ys[4] = {s.y0, s.y0, s.y1, s.y1}, xs[4] = {s.x0, s.x1, s.x0, s.x1};
      const float ws[4] = {s.w0, s.w1, s.w2, s.w3};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = ys[k] - a.pad, xx = xs[k] - a.pad;      // un-padded coordinates; the border is zero
        sIdx[e * 4 + k] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? yy * a.W + xx : -1;
        sW[e * 4 + k] = ws[k];
      }
    } else {
      const int* gi = a.gidx + ((size_t)pix * a.k2 + t) * km;
      const float* gwt = a.gw + ((size_t)pix * a.k2 + t) * km;
      for (int m = 0; m < km; ++m) { sIdx[e * km + m] = gi[m]; sW[e * km + m] = gwt[m]; }
    }
  }
  __syncthreads();
  const size_t sbase = (size_t)b * HW * a.C;
  uint4* __restrict__ dst = reinterpret_cast<uint4*>(a.dst) + ((size_t)b * HW + pix0) * a.k2 * a.nq;
  const float* __restrict__ src = a.src;
  const int nitems = npix * a.k2 * a.nq;
  // U items per trip with all their source loads issued before the first blend: the trips of a thread are serial round trips
  // to L2 otherwise (measured: 2.1 TB/s of G written with one item in flight per thread)
  auto run = [&](auto kmc, auto s16) {
    constexpr int KM = decltype(kmc)::value;
    constexpr int U = KM <= 4 ? 3 : (decltype(s16)::value ? 2 : 1);
    constexpr bool S16 = decltype(s16)::value;
    for (int it0 = threadIdx.x; it0 < nitems; it0 += 256 * U) {
      float4 lo[U][KM], hi[U][KM];
      float w[U][KM];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = min(it0 + u * 256, nitems - 1);
        const int e = a.lognq >= 0 ? it >> a.lognq : it / a.nq, q = it - e * a.nq;
#pragma unroll
        for (int m = 0; m < KM; ++m) {
          const int sp = sIdx[e * KM + m];
          w[u][m] = sp >= 0 ? sW[e * KM + m] : 0.f;
          const size_t eo = sbase + (size_t)(sp >= 0 ? sp : 0) * a.C + q * 8;
          if (S16) {
            const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(src) + eo);
            lo[u][m] = __builtin_bit_cast(float4, t);
          } else {
            lo[u][m] = *reinterpret_cast<const float4*>(src + eo);
            hi[u][m] = *reinterpret_cast<const float4*>(src + eo + 4);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
        for (int m = 0; m < KM; ++m) {
          float x8[8];
          if (S16) g16_unpack8(__builtin_bit_cast(uint4, lo[u][m]), x8);
          else { x8[0] = lo[u][m].x; x8[1] = lo[u][m].y; x8[2] = lo[u][m].z; x8[3] = lo[u][m].w;
                 x8[4] = hi[u][m].x; x8[5] = hi[u][m].y; x8[6] = hi[u][m].z; x8[7] = hi[u][m].w; }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += w[u][m] * x8[j];
        }
        uint4 h8, l8;
        pack8<false>(v, h8, l8);
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        if (it0 + u * 256 < nitems)
          __builtin_nontemporal_store(u32x4_t{h8.x, h8.y, h8.z, h8.w}, reinterpret_cast<u32x4_t*>(dst) + it0 + u * 256);   // (G is read back much later: keep it out of the way of the source rows in L2)
      }
    }
  };
  if (km == 4) { if (a.src_bf16) run(std::integral_constant<int, 4>{}, std::true_type{}); else run(std::integral_constant<int, 4>{}, std::false_type{}); }
  else if (km == 8) { if (a.src_bf16) run(std::integral_constant<int, 8>{}, std::true_type{}); else run(std::integral_constant<int, 8>{}, std::false_type{}); }
  else {
    for (int it = threadIdx.x; it < nitems; it += 256) {      // other table widths: the plain loop
      const int e = a.lognq >= 0 ? it >> a.lognq : it / a.nq, q = it - e * a.nq;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      for (int m = 0; m < km; ++m) {
        const int sp = sIdx[e * km + m];
        if (sp >= 0) {
          const float w = sW[e * km + m];
          float4 lo, hi;
          ld8any(src, a.src_bf16, sbase + (size_t)sp * a.C + q * 8, lo, hi);
          v[0] += w * lo.x; v[1] += w * lo.y; v[2] += w * lo.z; v[3] += w * lo.w;
          v[4] += w * hi.x; v[5] += w * hi.y; v[6] += w * hi.z; v[7] += w * hi.w;
        }
      }
      uint4 h8, l8;
      pack8<false>(v, h8, l8);
      dst[it] = h8;
    }
  }
}

// Taps per barrier round for a layer with C input channels on nwv waves when a thread may hold up to imax_max (pixel,
// 8-channel) items per round; returns whether two items per thread suffice (the smaller register variant of the forward).
// HDRSKY_DA_TPR caps it (1 = one tap per round, the layout before): tuning / test hook.
static bool da_taps_per_round(int C, int nwv, int k2, int imax_max, int* tpr) {
  const int nq = C / 8;
  int cap = imax_max * nwv;                 // 8-channel groups per round: imax * (64 nwv threads) / 64 pixels
  int t = cap / nq;
  if (t < 1) t = 1;
  if (t > k2) t = k2;
  { const int lim = hdrsky_hooks().da_tpr; if (lim >= 1 && t > lim) t = lim; }
  *tpr = t;
  return t * nq <= 2 * nwv;
}

// ---- kernel gradient with the region gather ---------------------------------------------------------------------
// dW[tap][ci][co] = sum over pixels of G[pixel][tap][ci] * dY[pixel][co] with G the layer's bilinear samples.  The weight-
// gradient kernel of conv_wgrad.hip (UP = 2) gathers from global memory, one tap's channels per workgroup and tile visit:
// every visit pays the dY tile, two barriers and a round trip to L2 for a handful of MFMAs (12 res layers: 1.5 ms per
// step, the two 7x7 layers 1.7 ms).  Here a workgroup stages the source rows of a GROUP of tiles of one sample once
// (as da_region_kernel) and loops rounds (tpr taps) outside, the group's tiles inside: per (round, tile) it gathers the A
// tile [64 pixels][tpr*C] from LDS, stages the dY tile [64][F block], and accumulates the round's dW fragments with
// K = pixels (transposing LDS reads, as conv_wgrad_kernel); after the group's tiles the round's block goes to the
// workgroup's partial slab ws[chunk = (sample, group)][k*k*C][F] with plain stores, and da_wgrad_reduce_kernel adds
// the chunks to dW in a fixed order (deterministic).  Rounds are dealt to `nsplit` workgroups per chunk.
typedef __attribute__((ext_vector_type(4))) short da_s16x4_t;
__device__ __forceinline__ uint2 da_lds_tr(const unsigned char* p) {
  const da_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) da_s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}

struct DaWgArgs {
  const float* x;
  const void* dy;
  const float* offs;
  const int* row_lo;
  float* ws;
  float* ws_db;
  int dy_bf16;
  int B, H, W, C, F, ksize, k2, pad, in_h, in_w;
  int tiles_x, grp_tiles, groups_x, src_rows, grp_rows;
  int tpr, nrounds, nsplit, nfblk, FB;
  int nq_sh, nqy_sh, cif_sh, cof_sh, nfr, w_sh, rt_cap;   // log2 of C/8, FB/8, C/16, FB/16; fragments per round; log2 W
  int RA, RY, y_off, tb_off, so_off, reg_off;   // LDS row strides / byte offsets
};

template <int NFR>
__global__ void __launch_bounds__(512) da_wgrad_region_kernel(const DaWgArgs a) {
  constexpr int TM = 64, NT = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float s_bias[8][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p = lane & 3, kq = g, lr = lane & 15;
  const int nq = a.C >> 3, nqr = nq * a.tpr, nqy = a.FB >> 3;
  unsigned char* sA = smem;                                  // [2][64 rows of RA bytes]
  unsigned char* sY = smem + a.y_off;                         // [2][64 rows of RY bytes]
  unsigned char* sTb = smem + a.tb_off;                       // [2] x { u16 idx[nent][4], float w[nent][4] }
  int4* sRT = reinterpret_cast<int4*>(smem + a.so_off);       // row halves of the sample positions of the group's rows
  float* sOx = reinterpret_cast<float*>(smem + a.so_off + a.rt_cap * 16);   // (as da_region_kernel), column offsets, tap % k
  int* sKx = reinterpret_cast<int*>(smem + a.so_off + a.rt_cap * 20);
  const uint4* sReg = reinterpret_cast<const uint4*>(smem + a.reg_off);
  const int nent = TM * a.tpr, tb_bytes = nent * 24;

  int bid = blockIdx.x;
  const int fb = bid % a.nfblk; bid /= a.nfblk;
  const int split = bid % a.nsplit; bid /= a.nsplit;
  const int grp = bid % a.groups_x, b = bid / a.groups_x;
  const int chunk = b * a.groups_x + grp;
  const int npix = a.H * a.W;
  const int ylo = a.row_lo[grp];
  const int nrows = min(a.src_rows, a.H - ylo);
  const int regpix = nrows * a.W;
  const int tile0 = grp * a.grp_tiles, ntl = min(a.grp_tiles, a.tiles_x - tile0);
  const int grow0 = (tile0 * TM) / a.W;                       // first image row of the group's output pixels
  const int f0 = fb * a.FB;

  // ---- group prologue: sampling offsets of its rows, source rows fp32 -> bf16 ----------------------------------
  {
    const int last = min((tile0 + ntl) * TM, npix) - 1;
    const int nr = last / a.W - grow0 + 1;
    for (int i = tid; i < nr * a.k2; i += NT) {
      const int r = i / a.k2, tn = i - r * a.k2;
      const float2 o = *reinterpret_cast<const float2*>(a.offs + ((size_t)grow0 * a.k2 + i) * 2);
      const TapAxis ty = da_tap_y((float)(grow0 + r + tn / a.ksize), o.x, a.in_h);
      const int r0 = ty.i0 - a.pad - ylo, r1 = ty.i1 - a.pad - ylo;
      sRT[i] = int4{(r0 >= 0 && r0 < nrows) ? r0 * a.W : -1, (r1 >= 0 && r1 < nrows) ? r1 * a.W : -1,
                    __builtin_bit_cast(int, ty.wa), __builtin_bit_cast(int, ty.wb)};
      sOx[i] = o.y;
    }
    for (int tn = tid; tn < a.k2; tn += NT) sKx[tn] = tn % a.ksize;
    const float* src = a.x + ((size_t)b * npix + (size_t)ylo * a.W) * a.C;
    uint4* dst = reinterpret_cast<uint4*>(smem + a.reg_off);
    const int nunits = regpix * nq;
    int u = tid;
    for (; u + 3 * NT < nunits; u += 4 * NT) {
      float4 lo[4], hi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* pp = src + (size_t)(u + k * NT) * 8;
        lo[k] = *reinterpret_cast<const float4*>(pp); hi[k] = *reinterpret_cast<const float4*>(pp + 4);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v[8] = {lo[k].x, lo[k].y, lo[k].z, lo[k].w, hi[k].x, hi[k].y, hi[k].z, hi[k].w};
        uint4 h8, l8;
        pack8<false>(v, h8, l8);
        dst[u + k * NT] = h8;
      }
    }
    for (; u < nunits; u += NT) {
      const float* pp = src + (size_t)u * 8;
      const float4 lo = *reinterpret_cast<const float4*>(pp), hi = *reinterpret_cast<const float4*>(pp + 4);
      const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      uint4 h8, l8;
      pack8<false>(v, h8, l8);
      dst[u] = h8;
    }
  }
  // this wave's dW fragments of a round: fr = k * 8 + wave -> (tap of the round ts, ci fragment i, co fragment j), i fastest
  int abase[NFR], ybase[NFR];
#pragma unroll
  for (int k = 0; k < NFR; ++k) {
    const int fr = k * 8 + wave;
    const int i = fr & ((1 << a.cif_sh) - 1), rest = fr >> a.cif_sh;
    const int j = rest & ((1 << a.cof_sh) - 1), ts = rest >> a.cof_sh;
    abase[k] = fr < a.nfr ? (ts * a.C + i * 16) * 2 : -1;
    ybase[k] = j * 32;
  }
  f32x4_t acc[NFR];
#pragma unroll
  for (int k = 0; k < NFR; ++k) acc[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  const int my_rounds = (a.nrounds - split + a.nsplit - 1) / a.nsplit;     // rounds split, split + nsplit, ...
  const int niter = my_rounds * ntl;
  const bool bias_wg = split == 0 && a.ws_db != nullptr;                    // sums dY over the group's tiles in its first round

  auto table_write = [&](int rnd, int tile, int slot) {
    unsigned short* ti = reinterpret_cast<unsigned short*>(sTb + slot * tb_bytes);
    float* tw = reinterpret_cast<float*>(sTb + slot * tb_bytes + nent * 8);
    const int p0 = tile * TM;
    for (int e = tid; e < nent; e += NT) {
      const int m = e & (TM - 1), tsub = e >> 6;
      const bool tap_ok = rnd * a.tpr + tsub < a.k2;
      const int tn = tap_ok ? rnd * a.tpr + tsub : a.k2 - 1;
      const bool live = p0 + m < npix;
      const int pix = live ? p0 + m : p0;
      const int ri = ((pix >> a.w_sh) - grow0) * a.k2 + tn, ox = pix & (a.W - 1);
      const int4 rt = sRT[ri];
      const TapAxis tx = da_tap_x((float)(ox + sKx[tn]), sOx[ri], a.in_w);
      const float wya = __builtin_bit_cast(float, rt.z), wyb = __builtin_bit_cast(float, rt.w);
      const int rys[4] = {rt.x, rt.x, rt.y, rt.y}, xs[4] = {tx.i0 - a.pad, tx.i1 - a.pad, tx.i0 - a.pad, tx.i1 - a.pad};
      const float ws4[4] = {wya * tx.wa, wya * tx.wb, wyb * tx.wa, wyb * tx.wb};
      unsigned short oi[4];
      float ow[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool in = live && tap_ok && rys[c] >= 0 && xs[c] >= 0 && xs[c] < a.W;
        oi[c] = (unsigned short)(in ? rys[c] + xs[c] : 0);
        ow[c] = in ? ws4[c] : 0.f;
      }
      *reinterpret_cast<uint2*>(ti + (size_t)e * 4) =
          uint2{(unsigned)oi[0] | ((unsigned)oi[1] << 16), (unsigned)oi[2] | ((unsigned)oi[3] << 16)};
      *reinterpret_cast<float4*>(tw + (size_t)e * 4) = float4{ow[0], ow[1], ow[2], ow[3]};
    }
  };
  // dY tile of `tile` -> registers (fp32 values), two (pixel, 8-channel) items per thread at most
  float yr[2][8];
  auto load_dy = [&](int tile) {
    const int p0 = tile * TM;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = k * NT + tid;
      if (idx < TM * nqy) {
        const int m = idx >> a.nqy_sh, qc = idx & (nqy - 1);
        const bool live = p0 + m < npix;
        const size_t eo = ((size_t)b * npix + (live ? p0 + m : 0)) * a.F + f0 + qc * 8;
        if (a.dy_bf16) {
          const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a.dy) + eo);
          const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            yr[k][2 * j] = live ? __builtin_bit_cast(float, w4[j] << 16) : 0.f;
            yr[k][2 * j + 1] = live ? __builtin_bit_cast(float, w4[j] & 0xffff0000u) : 0.f;
          }
        } else {
          const float* src = reinterpret_cast<const float*>(a.dy) + eo;
          const float4 va = *reinterpret_cast<const float4*>(src), vb = *reinterpret_cast<const float4*>(src + 4);
          yr[k][0] = live ? va.x : 0.f; yr[k][1] = live ? va.y : 0.f; yr[k][2] = live ? va.z : 0.f; yr[k][3] = live ? va.w : 0.f;
          yr[k][4] = live ? vb.x : 0.f; yr[k][5] = live ? vb.y : 0.f; yr[k][6] = live ? vb.z : 0.f; yr[k][7] = live ? vb.w : 0.f;
        }
      }
    }
  };

  __syncthreads();                               // offsets + region staged
  if (niter > 0) { table_write(split, tile0, 0); load_dy(tile0); }
  __syncthreads();
  int rr = 0, tl = 0;                            // iteration -> (round index of this workgroup, tile of the group)
  for (int it = 0; it < niter; ++it) {
    const int rnd = split + rr * a.nsplit, tile = tile0 + tl;
    const int nrr = tl + 1 == ntl ? rr + 1 : rr, ntl1 = tl + 1 == ntl ? 0 : tl + 1;   // the next iteration
    unsigned char* bufA = sA + (it & 1) * TM * a.RA;
    unsigned char* bufY = sY + (it & 1) * TM * a.RY;
    const unsigned short* ti = reinterpret_cast<const unsigned short*>(sTb + (it & 1) * tb_bytes);
    const float* tw = reinterpret_cast<const float*>(sTb + (it & 1) * tb_bytes + nent * 8);
    // ---- A tile: bilinear samples of tpr taps, pixel-major rows --------------------------------------------------
#pragma unroll 2
    for (int i = tid; i < TM * nqr; i += NT) {
      const int e = i >> a.nq_sh, q = i & (nq - 1);
      const int m = e & (TM - 1), qr = (e >> 6) * nq + q;
      const uint2 pi = *reinterpret_cast<const uint2*>(ti + (size_t)e * 4);
      const float4 pw = *reinterpret_cast<const float4*>(tw + (size_t)e * 4);
      const int oi[4] = {(int)(pi.x & 0xffffu), (int)(pi.x >> 16), (int)(pi.y & 0xffffu), (int)(pi.y >> 16)};
      const float ow[4] = {pw.x, pw.y, pw.z, pw.w};
      uint4 src[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) src[c] = sReg[oi[c] * nq + q];
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned w4[4] = {src[c].x, src[c].y, src[c].z, src[c].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[2 * j] += ow[c] * __builtin_bit_cast(float, w4[j] << 16);
          v[2 * j + 1] += ow[c] * __builtin_bit_cast(float, w4[j] & 0xffff0000u);
        }
      }
      uint4 h8, l8;
      pack8<false>(v, h8, l8);
      *reinterpret_cast<uint4*>(bufA + (size_t)m * a.RA + qr * 16) = h8;
    }
    // ---- dY tile (loaded an iteration ago) ------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = k * NT + tid;
      if (idx < TM * nqy) {
        const int m = idx >> a.nqy_sh, qc = idx & (nqy - 1);
        if (bias_wg && rr == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[j] += yr[k][j];
        }
        uint4 h8, l8;
        pack8<false>(yr[k], h8, l8);
        *reinterpret_cast<uint4*>(bufY + (size_t)m * a.RY + qc * 16) = h8;
      }
    }
    if (it + 1 < niter) table_write(split + nrr * a.nsplit, tile0 + ntl1, (it + 1) & 1);
    __syncthreads();
    if (it + 1 < niter) load_dy(tile0 + ntl1);
    // ---- MFMA: K = the tile's 64 pixels ----------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int m = r * 32 + g * 8 + q4;
#pragma unroll
      for (int k = 0; k < NFR; ++k) {
        if (abase[k] >= 0) {    // wave-uniform
          const unsigned char* ad = bufA + (size_t)m * a.RA + abase[k] + p * 8;
          const uint2 v0 = da_lds_tr(ad), v1 = da_lds_tr(ad + 4 * a.RA);
          const unsigned char* yd = bufY + (size_t)m * a.RY + ybase[k] + p * 8;
          const uint2 w0 = da_lds_tr(yd), w1 = da_lds_tr(yd + 4 * a.RY);
          acc[k] = mfma16(uint4{v0.x, v0.y, v1.x, v1.y}, uint4{w0.x, w0.y, w1.x, w1.y}, acc[k]);
        }
      }
    }
    if (tl + 1 == ntl) {      // the group's tiles are through: this round's block of dW goes to the partial slab
#pragma unroll
      for (int k = 0; k < NFR; ++k) {
        if (abase[k] >= 0) {
          const int fr = k * 8 + wave;
          const int i = fr & ((1 << a.cif_sh) - 1), rest = fr >> a.cif_sh;
          const int j = rest & ((1 << a.cof_sh) - 1), ts = rest >> a.cof_sh;
          const int tap = rnd * a.tpr + ts;
          if (tap < a.k2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = tap * a.C + i * 16 + kq * 4 + e, col = f0 + j * 16 + lr;
              a.ws[((size_t)chunk * a.k2 * a.C + row) * a.F + col] = acc[k][e];
            }
          }
        }
        acc[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    rr = nrr; tl = ntl1;
  }
  if (bias_wg) {      // column sums of dY over the group: lanes sharing a channel group, then the eight waves
    for (int mask = 32; mask >= nqy; mask >>= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum[j] += __shfl_xor(bsum[j], mask);
    }
    __syncthreads();
    if (lane < nqy) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s_bias[wave][lane * 8 + j] = bsum[j];
    }
    __syncthreads();
    if (tid < a.FB) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += s_bias[w][tid];
      a.ws_db[(size_t)chunk * a.F + f0 + tid] = t;
    }
  }
}

// dw[i] += sum over chunks of ws[c][i], db likewise - always in the same order.  A block owns 256/S float4 of the slab
// and S chunk slices: slice q sums the chunks q, q+S, ... (independent 16-byte loads in flight), then slice 0 adds the S
// partial sums in slice order (one chain over hundreds of chunks per element would be latency bound).
__global__ void __launch_bounds__(256) da_wgrad_reduce_kernel(const float4* ws, int nchunks, size_t n4, float4* dw,
                                                               const float* ws_db, int F, float* db, int S) {
  __shared__ float4 sPart[15 * 64];
  const int per = 256 / S, col = threadIdx.x % per, slice = threadIdx.x / per;
  const size_t v = (size_t)blockIdx.x * per + col;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (v < n4) {
    const float4* src = ws + v;
    int c = slice;
    for (; c + 3 * S < nchunks; c += 4 * S) {
      const float4 q0 = src[(size_t)c * n4], q1 = src[(size_t)(c + S) * n4];
      const float4 q2 = src[(size_t)(c + 2 * S) * n4], q3 = src[(size_t)(c + 3 * S) * n4];
      acc.x += (q0.x + q1.x) + (q2.x + q3.x); acc.y += (q0.y + q1.y) + (q2.y + q3.y);
      acc.z += (q0.z + q1.z) + (q2.z + q3.z); acc.w += (q0.w + q1.w) + (q2.w + q3.w);
    }
    for (; c < nchunks; c += S) {
      const float4 q = src[(size_t)c * n4];
      acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
    }
  }
  if (slice > 0) sPart[(slice - 1) * per + col] = acc;
  __syncthreads();
  if (slice == 0 && v < n4) {
    for (int k = 0; k < S - 1; ++k) { const float4 q = sPart[k * per + col]; acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
    float4 d = dw[v];
    d.x += acc.x; d.y += acc.y; d.z += acc.z; d.w += acc.w;
    dw[v] = d;
  }
  if (blockIdx.x == 0 && db != nullptr) {
    for (int co = threadIdx.x; co < F; co += 256) {
      float t = 0.f;
      for (int c = 0; c < nchunks; ++c) t += ws_db[(size_t)c * F + co];
      db[co] += t;
    }
  }
}

static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

// Plans the kernel above for a layer; returns the dynamic LDS bytes (0 = not supported here) and fills the geometry.
static int da_wgrad_plan(DaWgArgs& a, const int* spans) {
  const int C = a.C, F = a.F;
  if (!spans || !a.row_lo || (C % 32) != 0 || (F % 16) != 0) return 0;
  const int nq = C / 8;
  if ((nq & (nq - 1)) || C > 256 || (a.W & (a.W - 1))) return 0;
  a.w_sh = ilog2(a.W);
  a.FB = F > 128 ? 128 : F;
  if ((F % a.FB) != 0 || (a.FB & (a.FB - 1))) return 0;
  a.nfblk = F / a.FB;
  a.nq_sh = ilog2(nq); a.nqy_sh = ilog2(a.FB / 8); a.cif_sh = ilog2(C / 16); a.cof_sh = ilog2(a.FB / 16);
  const int hook_level = hdrsky_hooks().da_wg_group;     // tuning hook: tiles per region of the kernel-gradient launch (-1: the largest that fits)
  const int* row_lo_base = a.row_lo;
  for (int level = 4; level >= 0; --level) {      // the largest group that fits: fewest partial slabs
    const int G = 1 << level, groups = cdiv(a.tiles_x, G), src_rows = spans[level];
    if (hook_level >= 0 && level != hook_level) continue;
    if (level > 0 && G >= 2 * a.tiles_x) continue;                  // (no point in groups beyond the sample)
    if (src_rows <= 0 || (size_t)src_rows * a.W > 65535) continue;
    const int region = src_rows * a.W * C * 2;
    const int grp_rows = (G * 64) / a.W + 2;
    const int so = roundup(grp_rows * a.k2 * 20 + a.k2 * 4, 16);
    for (int tpr = a.k2 < 8 ? a.k2 : 8; tpr >= 1; --tpr) {
      const int nfr = tpr * (C / 16) * (a.FB / 16);
      if (nfr > 16 * 8 || 64 * tpr > 2 * 512) continue;
      const int RA = tpr * C * 2 + 16, RY = a.FB * 2 + 16;
      const int abytes = 2 * 64 * RA, ybytes = 2 * 64 * RY, tb = 2 * 64 * tpr * 24;
      if (abytes + ybytes + tb + so + region > 156 * 1024) continue;   // + 4 KB of static LDS (bias sums)
      a.tpr = tpr; a.nrounds = cdiv(a.k2, tpr); a.nfr = nfr;
      a.RA = RA; a.RY = RY; a.y_off = abytes; a.tb_off = abytes + ybytes; a.so_off = a.tb_off + tb; a.reg_off = a.so_off + so;
      a.src_rows = src_rows; a.grp_tiles = G; a.groups_x = groups; a.grp_rows = grp_rows; a.rt_cap = grp_rows * a.k2;
      a.row_lo = row_lo_base + (size_t)level * a.tiles_x;
      int nsplit = cdiv(256, a.B * groups * a.nfblk);     // rounds dealt to enough workgroups to fill the chip, evenly
      if (nsplit > a.nrounds) nsplit = a.nrounds;
      if (nsplit < 1) nsplit = 1;
      a.nsplit = cdiv(a.nrounds, cdiv(a.nrounds, nsplit));
      return abytes + ybytes + tb + so + region;
    }
  }
  return 0;
}

static int da_gs_launch(bool scatter, const float* src, const float* offs, int B, int H, int W, int C, int ksize, float* dst,
                        void* stream) {
  if (!src || !offs || !dst || (ksize & 1) == 0 || (C & 7)) return HDRSKY_EINVAL;
  DaGsArgs a{};
  a.src = src; a.dst = dst; a.offs = offs; a.B = B; a.H = H; a.W = W; a.C = C; a.ksize = ksize; a.k2 = ksize * ksize;
  a.pad = ksize > 1 ? (ksize - 1) / 2 : 0;
  a.in_h = H + (ksize > 1 ? ksize - 1 : 0); a.in_w = W + (ksize > 1 ? ksize - 1 : 0);
  const size_t total = (size_t)B * H * W * a.k2 * (C / 8);
  size_t grid = (total + 255) / 256; if (grid > 65535) grid = 65535; if (grid < 1) grid = 1;
  if (scatter) hipLaunchKernelGGL(da_gather_scatter_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(da_gather_scatter_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

}  // namespace

extern "C" {

// distortion_aware_ops.conv2d.distortion (:198-270) evaluated in float32 in the reference's operation order.
// out: [h][k*k][2] floats (y, x).  [host]
int hdrsky_da_offsets(int h, int w, int ksize, int dilation_rate, int skydome, float* out) {
  if (!out || h <= 0 || w <= 0 || ksize <= 0 || (ksize & 1) == 0) return HDRSKY_EINVAL;
  const float pi = (float)M_PI;
  const int n = ksize / 2, middle = n * (ksize + 1), k2 = ksize * ksize;
  const float unit_w = (float)(2.0 * M_PI) / (float)w;
  const float unit_h = pi / (float)(skydome ? h * 2 : h);
  const float rho = tanf(unit_w) * (float)dilation_rate;
  const int xc = (int)(w * 0.5);
  for (int y = 0; y < h; ++y) {
    const float theta = (float)((double)xc - 0.5 * (double)w) * unit_w;
    const float phi = skydome ? (float)(h - y) * unit_h : (float)((double)h * 0.5 - (double)y) * unit_h;
    const float pu[3] = {cosf(phi) * cosf(theta), sinf(phi), cosf(phi) * sinf(theta)};
    // t_x = cross((0,1,0), p_u), t_y = cross(p_u, t_x)   (not normalised)
    const float tx[3] = {1.f * pu[2] - 0.f * pu[1], 0.f * pu[0] - 0.f * pu[2], 0.f * pu[1] - 1.f * pu[0]};
    const float ty[3] = {pu[1] * tx[2] - pu[2] * tx[1], pu[2] * tx[0] - pu[0] * tx[2], pu[0] * tx[1] - pu[1] * tx[0]};
    float ky[64 * 2], kx[64 * 2];
    if (k2 > 128) return HDRSKY_EUNSUPPORTED;
    int i = 0;
    for (int gy = n; gy >= -n; --gy)
      for (int gx = n; gx >= -n; --gx, ++i) {
        float ur[3];
        for (int d = 0; d < 3; ++d) ur[d] = pu[d] + rho * ((float)gx * tx[d] + (float)gy * ty[d]);
        float theta_r;
        if (ur[0] > 0.f) theta_r = atan2f(ur[2], ur[0]);
        else if (ur[0] < 0.f) theta_r = ur[2] >= 0.f ? atan2f(ur[2], ur[0]) + pi : atan2f(ur[2], ur[0]) - pi;
        else if (ur[2] > 0.f) theta_r = pi * 0.5f;
        else if (ur[2] < 0.f) theta_r = -pi * 0.5f;
        else return HDRSKY_EINVAL;  // "undefined coordinates"
        const float phi_r = asinf(ur[1]);
        kx[i] = (theta_r / pi + 1.f) * 0.5f * (float)w;
        ky[i] = skydome ? (1.f - (2.f * phi_r) / pi) * (float)h : (0.5f - phi_r / pi) * (float)h;
      }
    for (i = 0; i < k2; ++i) {
      out[(y * k2 + i) * 2 + 0] = ky[i] - ky[middle];
      out[(y * k2 + i) * 2 + 1] = kx[i] - kx[middle];
    }
  }
  return HDRSKY_OK;
}

// y[B,H,W,Cout] = DA-conv(x[B,H,W,Cin]) + bias; weights packed with hdrsky_conv_pack_weights(w, k, k, Cin, Cout, 0, ..)
// from the reference's [k*k*Cin, Cout] kernel (same memory order as HWIO); offs = device copy of hdrsky_da_offsets.
int hdrsky_da_conv_stats_nparts(int H, int W) { return (H > 0 && W > 0) ? cdiv(H * W, 64) : 0; }

int hdrsky_da_conv2d_fwd(const float* x, const void* w_hi, const void* w_lo, const float* bias, const float* offs,
                         const int* row_lo, const int* spans, int B, int H, int W, int Cin, int Cout, int ksize, int compute,
                         float* y, float* stats_part, void* stream) {
  if (!x || !w_hi || !offs || !y || (ksize & 1) == 0) return HDRSKY_EINVAL;
  if ((Cin % 32) != 0) return HDRSKY_EUNSUPPORTED;
  const bool precise = compute == HDRSKY_BF16X3;
  if (precise && !w_lo) return HDRSKY_EINVAL;
  DaArgs a{};
  a.x = x; a.whi = (const uint4*)w_hi; a.wlo = (const uint4*)w_lo; a.bias = bias; a.offs = offs; a.y = y; a.stats = stats_part;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Npad = roundup(Cout, 64);
  a.ksize = ksize; a.k2 = ksize * ksize;
  // conv2d._pad_input (:125-150) for stride 1: pad (k-1)//2 before, rest after, when k > 1
  a.pad = ksize > 1 ? (ksize - 1) / 2 : 0;
  a.in_h = H + (ksize > 1 ? ksize - 1 : 0); a.in_w = W + (ksize > 1 ? ksize - 1 : 0);
  // workgroup: 64 pixels x 128 filters (8 waves) when the layer has more than 64 filters, else 64 filters (4 waves)
  const int nwv = Cout > 64 ? 8 : 4;
  a.cin32 = Cin / 32; a.nblocks = cdiv(Cout, nwv * 16); a.tiles_x = cdiv(H * W, 64);
  a.row_lo = row_lo; a.nparts = cdiv(H * W, 64);
  if (!precise) {    // the source rows of a tile staged once in LDS (da_region_kernel) when the caller knows their span
    a.nblocks = cdiv(Cout, 128);
    if (const int lds_r = da_region_plan(a, Cin, 8, 4, spans, B)) return da_region_launch<8, 4>(a, B * a.groups_x * a.nblocks, lds_r, stream);
    if (da_region_forced()) return HDRSKY_EUNSUPPORTED;
    a.row_lo = nullptr;
    a.nblocks = cdiv(Cout, nwv * 16);
  }
  if (64 * (Cin / 8) > 4 * nwv * 64) return HDRSKY_EUNSUPPORTED;   // register prefetch budget (Cin <= 128 / 256)
  // taps per round: as many as the per-round capacity (IMAX items per thread = 8 * IMAX * nwv channels) holds
  const bool imax2 = da_taps_per_round(Cin, nwv, a.k2, 4, &a.tpr);
  a.nrounds = cdiv(a.k2, a.tpr);
  const int lds = 2 * a.tpr * (Cin / 8) * 65 * 16 * (precise ? 2 : 1);
  if (lds > 152 * 1024) return HDRSKY_EUNSUPPORTED;          // + 5 KB of static LDS (the tile's offset table)
  a.tab_off = lds;
  a.use_tab = (lds + 64 * a.k2 * 32 <= 152 * 1024) ? 1 : 0;      // sample table: 32 B per (pixel, tap); 7x7 in BF16X3 does not fit
  if (hdrsky_hooks().da_tab == 0) a.use_tab = 0;               // switch (tests: the table-free path)
  if ((64 / W + 2) * a.k2 > 5 * 128) return HDRSKY_EUNSUPPORTED;   // rows a 64-pixel tile spans x taps: the staged offset table
  const int lds_launch = lds + (a.use_tab ? 64 * a.k2 * 32 : 0);
  const int grid = B * a.tiles_x * a.nblocks;
#define HDRSKY_DA_LAUNCH_(PREC_, NWV_, IMAX_)                                                                      \
  {                                                                                                               \
    auto k = da_conv_kernel<PREC_, NWV_, IMAX_, 4>;                                                                \
    static std::atomic<bool> set{false};                                                                                      \
    if (!set) {                                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                              152 * 1024) != hipSuccess) return HDRSKY_ELAUNCH;                                   \
      set = true;                                                                                                 \
    }                                                                                                             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(NWV_ * 64), lds_launch, (hipStream_t)stream, a);                              \
  }
#define HDRSKY_DA_LAUNCH(PREC_, NWV_) { if (imax2) HDRSKY_DA_LAUNCH_(PREC_, NWV_, 2) else HDRSKY_DA_LAUNCH_(PREC_, NWV_, 4) }
  if (precise) { if (nwv == 8) HDRSKY_DA_LAUNCH(true, 8) else HDRSKY_DA_LAUNCH(true, 4) }
  else { if (nwv == 8) HDRSKY_DA_LAUNCH(false, 8) else HDRSKY_DA_LAUNCH(false, 4) }
#undef HDRSKY_DA_LAUNCH
#undef HDRSKY_DA_LAUNCH_
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

// [host] The forward's sample table for an H x W map: for every (pixel p = oy*W + ox, tap t) the four bilinear corners as
// source pixel indices (row-major, -1 = in the zero padding) and weights - the arithmetic of da_tap (float32, reference
// order), i.e. exactly what the kernels gather.  idx / w: [H*W][k*k][4].  Its transpose (per target pixel: which
// (pixel, tap) samples read it, with which weight) is the table hdrsky_da_conv2d_dgrad wants; kernels.da_transpose_table
// builds it.
int hdrsky_da_sample_table(const float* offs, int H, int W, int ksize, int* idx, float* w) {
  if (!offs || !idx || !w || H <= 0 || W <= 0 || (ksize & 1) == 0) return HDRSKY_EINVAL;
  const int k2 = ksize * ksize, pad = ksize > 1 ? (ksize - 1) / 2 : 0;
  const int in_h = H + (ksize > 1 ? ksize - 1 : 0), in_w = W + (ksize > 1 ? ksize - 1 : 0);
  for (int oy = 0; oy < H; ++oy)
    for (int ox = 0; ox < W; ++ox)
      for (int t = 0; t < k2; ++t) {
        const Tap4 s = da_tap((float)(oy + t / ksize), (float)(ox + t % ksize), offs[(oy * k2 + t) * 2], offs[(oy * k2 + t) * 2 + 1],
                              in_h, in_w);
        const int ys[4] = {s.y0, s.y0, s.y1, s.y1}, xs[4] = {s.x0, s.x1, s.x0, s.x1};
        const float ws[4] = {s.w0, s.w1, s.w2, s.w3};
        for (int k = 0; k < 4; ++k) {
          const int yy = ys[k] - pad, xx = xs[k] - pad;
          const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
          const size_t e = (((size_t)oy * W + ox) * k2 + t) * 4 + k;
          idx[e] = in ? yy * W + xx : -1;
          w[e] = in ? ws[k] : 0.f;
        }
      }
  return HDRSKY_OK;
}

// Data gradient of the distortion-aware conv without the k*k x tensor and without atomics:
//   dx[q][c] = sum_t sum_f ( sum_{(p,w) in L(q,t)} w * dy[p][f] ) * W[t][c][f]
// i.e. the SAME kernel as the forward - a weighted gather into the LDS tile per tap, then MFMA against the tap's filter -
// with (a) the transposed sample table L (gidx / gw [H*W][k*k][8]: up to 8 source pixels per (target pixel, tap), tap
// order of the packed filter) and (b) the transpose_flip image of the kernel (Cin = the layer's filters, Cout = its input
// channels; its taps are flipped, which the table's tap order accounts for).  Deterministic.
int hdrsky_da_conv2d_dgrad(const float* dy, const void* wT_hi, const void* wT_lo, const int* gidx, const float* gw,
                           const int* row_lo, const int* spans, int B, int H, int W, int F, int C, int ksize, int compute, float* dx,
                           void* stream) {
  if (!dy || !wT_hi || !gidx || !gw || !dx || (ksize & 1) == 0) return HDRSKY_EINVAL;
  if ((F % 32) != 0) return HDRSKY_EUNSUPPORTED;
  const bool precise = compute == HDRSKY_BF16X3;
  if (precise && !wT_lo) return HDRSKY_EINVAL;
  DaArgs a{};
  a.x = dy; a.whi = (const uint4*)wT_hi; a.wlo = (const uint4*)wT_lo; a.y = dx; a.gidx = gidx; a.gw = gw;
  a.B = B; a.H = H; a.W = W; a.Cin = F; a.Cout = C; a.Npad = roundup(C, 64);
  a.ksize = ksize; a.k2 = ksize * ksize;
  a.pad = ksize > 1 ? (ksize - 1) / 2 : 0;
  a.in_h = H + (ksize > 1 ? ksize - 1 : 0); a.in_w = W + (ksize > 1 ? ksize - 1 : 0);
  // 8 waves when dx has more than 64 channels - or when dy has (two (pixel, 8-channel) items of 8 sources each per thread
  // is the register budget: 64 pixels x F/8 items need F/16 waves)
  const int nwv = (C > 64 || F > 64) ? 8 : 4;
  a.cin32 = F / 32; a.nblocks = cdiv(C, nwv * 16); a.tiles_x = cdiv(H * W, 64);
  a.row_lo = row_lo;
  if (!precise) {
    a.nblocks = cdiv(C, 128);
    if (const int lds_r = da_region_plan(a, F, 8, 8, spans, B)) return da_region_launch<8, 8>(a, B * a.groups_x * a.nblocks, lds_r, stream);
    if (da_region_forced()) return HDRSKY_EUNSUPPORTED;
    a.row_lo = nullptr;
    a.nblocks = cdiv(C, nwv * 16);
  }
  if (64 * (F / 8) > 2 * nwv * 64) return HDRSKY_EUNSUPPORTED;
  da_taps_per_round(F, nwv, a.k2, 2, &a.tpr);
  a.nrounds = cdiv(a.k2, a.tpr);
  const int lds = 2 * a.tpr * (F / 8) * 65 * 16 * (precise ? 2 : 1);
  if (lds > 152 * 1024) return HDRSKY_EUNSUPPORTED;
  if ((64 / W + 2) * a.k2 > 5 * 128) return HDRSKY_EUNSUPPORTED;
  a.tab_off = lds; a.use_tab = 0;
  const int grid = B * a.tiles_x * a.nblocks;
#define HDRSKY_DAG_LAUNCH(PREC_, NWV_)                                                                             \
  {                                                                                                               \
    auto k = da_conv_kernel<PREC_, NWV_, 2, 8>;                                                                    \
    static std::atomic<bool> set{false};                                                                                      \
    if (!set) {                                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                              152 * 1024) != hipSuccess) return HDRSKY_ELAUNCH;                                   \
      set = true;                                                                                                 \
    }                                                                                                             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(NWV_ * 64), lds, (hipStream_t)stream, a);                                \
  }
  if (precise) { if (nwv == 8) HDRSKY_DAG_LAUNCH(true, 8) else HDRSKY_DAG_LAUNCH(true, 4) }
  else { if (nwv == 8) HDRSKY_DAG_LAUNCH(false, 8) else HDRSKY_DAG_LAUNCH(false, 4) }
#undef HDRSKY_DAG_LAUNCH
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

// G[B,H,W,k*k*C] = bilinear gather of x[B,H,W,C] (the operand of the layer's matmul, distortion_aware_ops.py:62-113)
int hdrsky_da_gather(const float* x, const float* offs, int B, int H, int W, int C, int ksize, float* G, void* stream) {
  return da_gs_launch(false, x, offs, B, H, W, C, ksize, G, stream);
}

// G (bf16) [B,H,W,k*k*C]: the operand the matrix cores see, from the forward's corners (offs) or from a sample table
int hdrsky_da_gather_bf16(const void* x, int x_bf16, const float* offs, const int* gidx, const float* gw, int km, int B, int H,
                          int W, int C, int ksize, void* G, void* stream) {
  if (!x || !G || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (ksize & 1) == 0 || ksize > 7) return HDRSKY_EINVAL;
  if (!offs && (!gidx || !gw || km <= 0)) return HDRSKY_EINVAL;
  if ((size_t)H * W * ksize * ksize * (size_t)(km > 0 ? km : 1) >= ((size_t)1 << 31)) return HDRSKY_EUNSUPPORTED;
  DaG16Args a{};
  a.src = (const float*)x; a.src_bf16 = x_bf16; a.dst = (unsigned short*)G; a.offs = offs; a.gidx = gidx; a.gw = gw; a.km = km;
  a.B = B; a.H = H; a.W = W; a.C = C; a.ksize = ksize; a.k2 = ksize * ksize;
  a.pad = ksize > 1 ? (ksize - 1) / 2 : 0;
  a.in_h = H + (ksize > 1 ? ksize - 1 : 0); a.in_w = W + (ksize > 1 ? ksize - 1 : 0);
  if (!offs && km > DA_G16_KM) return HDRSKY_EUNSUPPORTED;
  a.nq = C >> 3;
  a.lognq = -1;
  for (int l = 0; l < 12; ++l) if ((1 << l) == a.nq) a.lognq = l;
  // pixels per workgroup: ~2304 items (9 per thread), at most what 32 KB of position entries hold, and a y grid dimension per sample
  int pixb = 2304 / (a.k2 * a.nq);
  pixb = pixb < 1 ? 1 : (pixb > 64 ? 64 : pixb);
  const int kme = offs ? 4 : km;
  while (pixb > 1 && pixb * a.k2 * kme * 8 > 32 * 1024) pixb >>= 1;
  a.pixb = pixb;
  if (B > 65535) return HDRSKY_EUNSUPPORTED;
  hipLaunchKernelGGL(da_gather_bf16_kernel, dim3(cdiv(H * W, pixb), B), dim3(256), pixb * a.k2 * kme * 8, (hipStream_t)stream, a);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

// dx[B,H,W,C] += transpose of the gather applied to dG[B,H,W,k*k*C] (fp32 atomics: zero dx first)
int hdrsky_da_scatter(const float* dG, const float* offs, int B, int H, int W, int C, int ksize, float* dx, void* stream) {
  return da_gs_launch(true, dG, offs, B, H, W, C, ksize, dx, stream);
}

// Kernel gradient of the distortion-aware conv with the region gather (BF16 mode): dw [k*k*C, F] += G(x)^T dy,
// db [F] += column sums of dy (db may be NULL).  offs = device offsets, row_lo / spans as for hdrsky_da_conv2d_fwd.
// ws: hdrsky_da_conv2d_wgrad_ws_bytes(...) bytes of device scratch (partial slabs per (sample, tile group)).
// Returns HDRSKY_EUNSUPPORTED when the layer does not fit (callers then use hdrsky_conv2d_wgrad_multi's da_* job).
static int da_wgrad_setup(DaWgArgs& a, const int* row_lo, const int* spans, int B, int H, int W, int C, int F, int ksize) {
  a = DaWgArgs{};
  a.row_lo = row_lo; a.B = B; a.H = H; a.W = W; a.C = C; a.F = F; a.ksize = ksize; a.k2 = ksize * ksize;
  a.pad = ksize > 1 ? (ksize - 1) / 2 : 0;
  a.in_h = H + (ksize > 1 ? ksize - 1 : 0); a.in_w = W + (ksize > 1 ? ksize - 1 : 0);
  a.tiles_x = cdiv(H * W, 64);
  if (B <= 0 || H <= 0 || W <= 0 || (ksize & 1) == 0 || ksize > 7) return 0;
  return da_wgrad_plan(a, spans);
}

size_t hdrsky_da_conv2d_wgrad_ws_bytes(const int* row_lo, const int* spans, int B, int H, int W, int C, int F, int ksize) {
  DaWgArgs a;
  if (!da_wgrad_setup(a, row_lo, spans, B, H, W, C, F, ksize)) return 0;
  const size_t chunks = (size_t)B * a.groups_x;
  return (chunks * a.k2 * C * F + chunks * F) * sizeof(float);
}

int hdrsky_da_conv2d_wgrad(const float* x, const void* dy, int dy_bf16, const float* offs, const int* row_lo, const int* spans,
                           int B, int H, int W, int C, int F, int ksize, float* dw, float* db, void* ws, size_t ws_bytes,
                           void* stream) {
  if (!x || !dy || !offs || !dw || !ws) return HDRSKY_EINVAL;
  DaWgArgs a;
  const int lds = da_wgrad_setup(a, row_lo, spans, B, H, W, C, F, ksize);
  if (!lds) return HDRSKY_EUNSUPPORTED;
  const size_t chunks = (size_t)B * a.groups_x, slab = (size_t)a.k2 * C * F;
  if (ws_bytes < (chunks * slab + chunks * F) * sizeof(float)) return HDRSKY_EINVAL;
  a.x = x; a.dy = dy; a.dy_bf16 = dy_bf16; a.offs = offs;
  a.ws = (float*)ws; a.ws_db = db ? (float*)ws + chunks * slab : nullptr;
  const int grid = B * a.groups_x * a.nsplit * a.nfblk;
  const int nfr_wave = cdiv(a.nfr, 8);
#define HDRSKY_DAWG(NFR_)                                                                                          \
  {                                                                                                               \
    auto k = da_wgrad_region_kernel<NFR_>;                                                                        \
    static std::atomic<bool> set{false};                                                                                      \
    if (!set) {                                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                              156 * 1024) != hipSuccess) return HDRSKY_ELAUNCH;                                   \
      set = true;                                                                                                 \
    }                                                                                                             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, (hipStream_t)stream, a);                                    \
  }
  if (nfr_wave <= 4) HDRSKY_DAWG(4) else if (nfr_wave <= 8) HDRSKY_DAWG(8) else HDRSKY_DAWG(16)
#undef HDRSKY_DAWG
  HDRSKY_CHECK_LAUNCH();
  const size_t n4 = slab / 4;
  const int S = chunks >= 64 ? 16 : 4, per = 256 / S;
  hipLaunchKernelGGL(da_wgrad_reduce_kernel, dim3((unsigned)((n4 + per - 1) / per)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)ws, (int)chunks, n4, (float4*)dw, a.ws_db, F, db, S);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

}  // extern "C"
