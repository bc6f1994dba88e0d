// Per-CU load calibration: every workgroup (256 threads) pulls BYTES_PER_WG from a buffer with
// DEPTH independent 16-byte loads in flight per thread; reports cycles/WG and B/clk/CU.
// modes: shared = all WGs read the same region (weights-like); private = each WG its own region.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int DEPTH>
__global__ void __launch_bounds__(256) pull(const uint4* __restrict__ src, size_t wg_stride_u4, int iters,
                                            unsigned long long* stamps, float* sink) {
  const uint4* p = src + (size_t)blockIdx.x * wg_stride_u4 + threadIdx.x;
  unsigned acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    uint4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = p[(size_t)(it * DEPTH + d) * 256];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
  if (acc == 0x12345678u) sink[0] = 1.f;
}
template <int DEPTH>
void run(const char* name, const uint4* buf, size_t stride_u4, int bytes_per_wg, int nwg) {
  unsigned long long* st; float* sink;
  hipMalloc(&st, nwg * 16); hipMalloc(&sink, 4);
  int iters = bytes_per_wg / (256 * 16 * DEPTH);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(pull<DEPTH>, dim3(nwg), dim3(256), 0, 0, buf, stride_u4, iters, st, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nwg * 2);
  hipMemcpy(h.data(), st, nwg * 16, hipMemcpyDeviceToHost);
  std::vector<double> d;
  for (int i = 0; i < nwg; ++i) d.push_back((double)(h[2 * i + 1] - h[2 * i]));
  std::sort(d.begin(), d.end());
  double med = d[nwg / 2];
  printf("%-28s depth %2d  WGs %4d  %6d B/WG : median %8.0f cyc/WG  -> %6.1f B/clk/CU, %7.0f cyc per batch of %d loads\n",
         name, DEPTH, nwg, bytes_per_wg, med, bytes_per_wg / med, med / iters, DEPTH);
  hipFree(st); hipFree(sink);
}
int main() {
  const size_t total = 512ull << 20;
  uint4* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
  const int B = 64 * 1024;
  // all WGs read the same 64 KB (L2-resident, weights-like)
  run<1>("shared 64KB", buf, 0, B, 256); run<4>("shared 64KB", buf, 0, B, 256); run<8>("shared 64KB", buf, 0, B, 256); run<16>("shared 64KB", buf, 0, B, 256);
  // each WG its own 64 KB, total 16 MB (L2/MALL-resident after warm-up)
  run<1>("private 64KB (16MB total)", buf, B / 16, B, 256); run<4>("private 64KB (16MB total)", buf, B / 16, B, 256);
  run<8>("private 64KB (16MB total)", buf, B / 16, B, 256); run<16>("private 64KB (16MB total)", buf, B / 16, B, 256);
  // 4 WGs per CU
  run<8>("private 64KB x1024 WGs", buf, B / 16, B, 1024); run<8>("shared 64KB x1024 WGs", buf, 0, B, 1024);
  // HBM: each WG its own 1 MB stride (256 MB footprint... touches only 64KB each) cold-ish
  run<8>("private 64KB, 1MB stride", buf, (1 << 20) / 16, B, 256);
  return 0;
}
