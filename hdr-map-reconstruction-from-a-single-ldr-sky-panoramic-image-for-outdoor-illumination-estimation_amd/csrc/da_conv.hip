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

#include "common.h"

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
  int src_rows, tb_off, reg_off, nq_sh, grp_tiles, groups_x;
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
      float* dst = a.stats + ((size_t)(b * a.tiles_x + tile) * 2) * a.Cout + n;
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
template <int NWV, int KM, int CBMAX>
__global__ void __launch_bounds__(NWV * 64) da_region_kernel(const DaArgs a) {
  constexpr int TM = 64, NT = NWV * 64;
  constexpr int MAXROWS = 5, EMAX = 2;                          // CBMAX: 32-channel k-steps per round (register sets)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, lr = lane & 15;
  const int nq = a.Cin >> 3, nqr = nq * a.tpr, plane = TM + 1, buf_units = nqr * plane;
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
  const int ylo = a.row_lo[grp];
  const int nrows = min(a.src_rows, a.H - ylo);
  const int regpix = nrows * a.W;
  const int tile_end = min((grp + 1) * a.grp_tiles, a.tiles_x);

  f32x4_t acc[4];
  const bool wave_live = n0 + wave * 16 < a.Npad;     // waves beyond the filter image only help with the gather
  const int wcol = wave_live ? n0 + wave * 16 : 0;
  const uint4* wlh = a.whi + (size_t)kq * a.Npad + wcol + lr;

  __shared__ float s_off[MAXROWS * 2 * 128];
  // ---- the source rows of this group: a linear fp32 -> bf16 copy -------------------------------------------------
  {
    const float* src = a.x + ((size_t)b * npix + (size_t)ylo * a.W) * a.Cin;
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
  uint4 bch[CBMAX], bnh[CBMAX];
  const int ksteps = a.k2 * a.cin32, spr = a.tpr * a.cin32;
  const int nitems = TM * nqr;
  for (int tile = grp * a.grp_tiles; tile < tile_end; ++tile) {
  const int p0 = tile * TM;
  const int row0 = p0 / a.W;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) acc[mi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if constexpr (KM == 4) {
    const int nrow = min(p0 + TM - 1, npix - 1) / a.W - row0 + 1;
    for (int i = tid; i < nrow * a.k2 * 2; i += NT) s_off[i] = a.offs[(size_t)row0 * a.k2 * 2 + i];
    __syncthreads();     // s_off visible to the table writers
  }

  // ---- the per-round sample table ---------------------------------------------------------------------------------
  int pend_i[KM == 8 ? EMAX : 1][8];
  float pend_w[KM == 8 ? EMAX : 1][8];
  auto table_fetch = [&](int rnd) {            // KM = 8: global loads of round rnd's entries into registers
    if constexpr (KM == 8) {
#pragma unroll
      for (int k = 0; k < EMAX; ++k) {
        const int e = k * NT + tid;
        if (e < nent) {
          const int m = e % TM, tn = min(rnd * a.tpr + e / TM, a.k2 - 1);
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
        const int m = e % TM, tsub = e / TM;
        const bool tap_ok = rnd * a.tpr + tsub < a.k2;
        const int tn = tap_ok ? rnd * a.tpr + tsub : a.k2 - 1;
        const bool live = p0 + m < npix;
        unsigned short oi[KM];
        float ow[KM];
        if constexpr (KM == 4) {
          const int pix = p0 + m;
          const int oy = live ? pix / a.W : row0, ox = live ? pix % a.W : 0;
          const float off_y = s_off[((oy - row0) * a.k2 + tn) * 2], off_x = s_off[((oy - row0) * a.k2 + tn) * 2 + 1];
          const Tap4 s = da_tap((float)(oy + tn / a.ksize), (float)(ox + tn % a.ksize), off_y, off_x, a.in_h, a.in_w);
          const int ys[4] = {s.y0, s.y0, s.y1, s.y1}, xs[4] = {s.x0, s.x1, s.x0, s.x1};
          const float ws[4] = {s.w0, s.w1, s.w2, s.w3};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int yy = ys[c] - a.pad - ylo, xx = xs[c] - a.pad;     // region row / column; outside = zero padding
            const bool in = live && tap_ok && yy >= 0 && yy < nrows && ys[c] - a.pad < a.H && xx >= 0 && xx < a.W;
            oi[c] = (unsigned short)(in ? yy * a.W + xx : 0);
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
      if (cb < spr && rnd * spr + cb < ksteps) bnh[cb] = wlh[(size_t)((rnd * spr + cb) * 4) * a.Npad];
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
      // item order (tap of the round, pixel, 8-channel group): nq is a power of two and TM = 64, so no division - the
      // nq lanes of a (pixel, tap) share one table entry (LDS broadcast)
      const int e = i >> a.nq_sh, q = i & (nq - 1);
      const int m = e & (TM - 1), qr = (e >> 6) * nq + q;
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
        if (cb >= spr || t * spr + cb >= ksteps) break;
        const uint4 bh = bch[cb];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = mfma16(buf[(cb * 4 + kq) * plane + mi * 16 + lr], bh, acc[mi]);
      }
#pragma unroll
      for (int cb = 0; cb < CBMAX; ++cb) bch[cb] = bnh[cb];
    }
  }
  // ---- epilogue: + bias, store (as da_conv_kernel) --------------------------------------------------------------
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
  if (a.stats) {
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (kq == 0 && n < a.Cout) {
      float* dst = a.stats + ((size_t)(b * a.tiles_x + tile) * 2) * a.Cout + n;
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
  if ((64 / a.W + 2) * a.k2 > 5 * 128) return 0;
  int hook_level = -1;
  if (const char* e = getenv("HDRSKY_DA_REGION")) { if (atoi(e) == 0) return 0; }   // tuning / test hooks
  if (const char* e = getenv("HDRSKY_DA_GROUP")) hook_level = atoi(e);
  const int nq = C / 8, cin32 = C / 32, nt = nwv * 64;
  if (nq & (nq - 1)) return 0;                                  // the item indexing wants a power of two
  a.nq_sh = __builtin_ctz(nq);
  const int* row_lo_base = a.row_lo;
  for (int level = 4; level >= 0; --level) {
    const int G = 1 << level, groups = cdiv(a.tiles_x, G), src_rows = spans[level];
    if (level != (hook_level >= 0 ? hook_level : 0)) continue;   // measured: groups of tiles do not pay (see DESIGN), hook only
    if (src_rows <= 0 || (size_t)src_rows * a.W > 65535) continue;   // u16 region pixel indices
    const int region = src_rows * a.W * C * 2;
    for (int tpr = a.k2 < 8 ? a.k2 : 8; tpr >= 1; --tpr) {
      if (tpr * cin32 > 8 || 64 * tpr > 2 * nt) continue;     // filter prefetch registers, table entries per thread
      const int abytes = 2 * tpr * nq * 65 * 16, tb = 2 * 64 * tpr * km * 6;
      if (abytes + tb + region > 152 * 1024) continue;
      a.tpr = tpr; a.nrounds = cdiv(a.k2, tpr);
      a.tb_off = abytes; a.reg_off = abytes + tb; a.src_rows = src_rows;
      a.grp_tiles = G; a.groups_x = groups; a.row_lo = row_lo_base + (size_t)level * a.tiles_x;
      return abytes + tb + region;
    }
  }
  return 0;
}

// HDRSKY_DA_REGION=2: fail instead of falling back to the global-memory gather (tests: proves which kernel ran)
static bool da_region_forced() { const char* e = getenv("HDRSKY_DA_REGION"); return e && atoi(e) == 2; }

template <int NWV, int KM, int CBMAX>
static int da_region_launch_(const DaArgs& a, int grid, int lds, void* stream) {
  auto k = da_region_kernel<NWV, KM, CBMAX>;
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
      return HDRSKY_ELAUNCH;
    set = true;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NWV * 64), lds, (hipStream_t)stream, a);
  HDRSKY_CHECK_LAUNCH();
  return HDRSKY_OK;
}

template <int NWV, int KM>
static int da_region_launch(const DaArgs& a, int grid, int lds, void* stream) {
  return a.tpr * a.cin32 <= 4 ? da_region_launch_<NWV, KM, 4>(a, grid, lds, stream) : da_region_launch_<NWV, KM, 8>(a, grid, lds, stream);
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

// Taps per barrier round for a layer with C input channels on nwv waves when a thread may hold up to imax_max (pixel,
// 8-channel) items per round; returns whether two items per thread suffice (the smaller register variant of the forward).
// HDRSKY_DA_TPR caps it (1 = one tap per round, the layout before): tuning / test hook.
static bool da_taps_per_round(int C, int nwv, int k2, int imax_max, int* tpr) {
  const int nq = C / 8;
  int cap = imax_max * nwv;                 // 8-channel groups per round: imax * (64 nwv threads) / 64 pixels
  int t = cap / nq;
  if (t < 1) t = 1;
  if (t > k2) t = k2;
  if (const char* e = getenv("HDRSKY_DA_TPR")) { const int lim = atoi(e); if (lim >= 1 && t > lim) t = lim; }
  *tpr = t;
  return t * nq <= 2 * nwv;
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
  a.row_lo = row_lo;
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
  if (const char* e = getenv("HDRSKY_DA_TAB")) { if (atoi(e) == 0) a.use_tab = 0; }   // tuning / test hook
  if ((64 / W + 2) * a.k2 > 5 * 128) return HDRSKY_EUNSUPPORTED;   // rows a 64-pixel tile spans x taps: the staged offset table
  const int lds_launch = lds + (a.use_tab ? 64 * a.k2 * 32 : 0);
  const int grid = B * a.tiles_x * a.nblocks;
#define HDRSKY_DA_LAUNCH_(PREC_, NWV_, IMAX_)                                                                      \
  {                                                                                                               \
    auto k = da_conv_kernel<PREC_, NWV_, IMAX_, 4>;                                                                \
    static bool set = false;                                                                                      \
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
    static bool set = false;                                                                                      \
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

// dx[B,H,W,C] += transpose of the gather applied to dG[B,H,W,k*k*C] (fp32 atomics: zero dx first)
int hdrsky_da_scatter(const float* dG, const float* offs, int B, int H, int W, int C, int ksize, float* dx, void* stream) {
  return da_gs_launch(true, dG, offs, B, H, W, C, ksize, dx, stream);
}

}  // extern "C"
