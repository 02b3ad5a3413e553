// What does the memory system sustain for the traffic SHAPE of the 5x5 correlation kernels - a workgroup reads R contiguous
// bytes and writes W contiguous bytes, nothing else - and does it matter how many workgroups a CU holds or whether they are
// persistent?  (Ceiling for xcorr_cfg5_kernel: R = 19,600, W = 15,376 per 4 planes; xcorr_prod29_kernel: 13,456 / 10,000.)
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/ubench_stream.hip -o tools/experiments/ubench_stream && tools/experiments/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <type_traits>

// one-shot: group g = blockIdx.x; loads -> LDS -> barrier -> LDS -> stores (the structure of the correlation kernels)
typedef float f4n __attribute__((ext_vector_type(4)));
// NT: 1 = nontemporal loads, 2 = nontemporal stores, 3 = both (streaming hints: the data is touched once)
template <int R4, int W4, int NT = 0>
__global__ __launch_bounds__(256) void oneshot(const float4* __restrict__ in, float4* __restrict__ out, int groups, int lds_pad) {
  extern __shared__ float4 sm[];
  const int tid = threadIdx.x;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const float4* s = in + (size_t)g * R4;
    float4* d = out + (size_t)g * W4;
    constexpr int RI = (R4 + 255) / 256, WI = (W4 + 255) / 256;
    float4 r[RI];
#pragma unroll
    for (int q = 0; q < RI; ++q) {
      if constexpr (NT & 1) { const f4n v = __builtin_nontemporal_load(reinterpret_cast<const f4n*>(s) + min(tid + q * 256, R4 - 1)); r[q] = float4{v.x, v.y, v.z, v.w}; }
      else r[q] = s[min(tid + q * 256, R4 - 1)];
    }
#pragma unroll
    for (int q = 0; q < RI; ++q) if (tid + q * 256 < R4) sm[tid + q * 256] = r[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < WI; ++q) if (tid + q * 256 < W4) {
      const float4 v = sm[tid + q * 256];
      if constexpr (NT & 2) __builtin_nontemporal_store(f4n{v.x, v.y, v.z, v.w}, reinterpret_cast<f4n*>(d) + tid + q * 256);
      else d[tid + q * 256] = v;
    }
    __syncthreads();
  }
}

// straight copy, no LDS: the practical ceiling
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void read4(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  float4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = in[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  if (a.x == 1.2345f) out[0] = a;
}
__global__ __launch_bounds__(256) void write4(float4* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = float4{1, 2, 3, 4};
}

template <class F>
double timed(F f) {
  for (int i = 0; i < 20; ++i) f();
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, (double)ms / 20 * 1e3);
  }
  return best;
}

template <int R4, int W4>
void shape(const char* name, const float4* in, float4* out, int groups) {
  const double bytes = (double)groups * (R4 + W4) * 16;
  for (int lds_kb : {20, 27, 35, 40, 53, 80}) {   // LDS per workgroup decides how many a CU holds: 8, 5, 4, 4, 3, 2
    const int lds = lds_kb * 1024;
    hipFuncSetAttribute((const void*)oneshot<R4, W4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    double t = timed([&] { oneshot<R4, W4><<<groups, 256, lds>>>(in, out, groups, 0); });
    const int per_cu = 160 / lds_kb;
    double tp = timed([&] { oneshot<R4, W4><<<256 * per_cu, 256, lds>>>(in, out, groups, 0); });
    printf("%-28s LDS %2d KB (%d WG/CU): one-shot grid %.1f us = %.0f GB/s | persistent %d WGs %.1f us = %.0f GB/s\n", name, lds_kb, per_cu, t,
           bytes / t / 1e3, 256 * per_cu, tp, bytes / tp / 1e3);
  }
}

int main() {
  const int groups = 6 * 64 * 256 / 4;  // 6 problems x 16,384 planes / 4 planes per workgroup
  size_t n4 = (size_t)groups * 1225 + 4096;
  float4 *in, *out;
  hipMalloc(&in, n4 * 16); hipMalloc(&out, n4 * 16);
  hipMemset(in, 0, n4 * 16);
  for (int blocks : {2048, 4096, 8192}) {
    double t = timed([&] { copy4<<<blocks, 256>>>(in, out, n4); });
    printf("copy  %zu MB x2, %d WGs: %.1f us = %.0f GB/s (read + write)\n", n4 * 16 >> 20, blocks, t, 2.0 * n4 * 16 / t / 1e3);
  }
  { double t = timed([&] { read4<<<4096, 256>>>(in, out, n4); }); printf("read  only: %.1f us = %.0f GB/s\n", t, 1.0 * n4 * 16 / t / 1e3); }
  { double t = timed([&] { write4<<<4096, 256>>>(out, n4); }); printf("write only: %.1f us = %.0f GB/s\n", t, 1.0 * n4 * 16 / t / 1e3); }
  {   // streaming hints on the two production shapes at their kernels' occupancy (cfg5: 20 KB -> 8 WG/CU; prod29: 27 KB -> 5 WG/CU)
    auto nt = [&](auto NTc, const char* tag) {
      constexpr int N = decltype(NTc)::value;
      hipFuncSetAttribute((const void*)oneshot<1225, 961, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 20 * 1024);
      hipFuncSetAttribute((const void*)oneshot<841, 625, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 27 * 1024);
      const double t5 = timed([&] { oneshot<1225, 961, N><<<groups, 256, 20 * 1024>>>(in, out, groups, 0); });
      const double t9 = timed([&] { oneshot<841, 625, N><<<groups, 256, 27 * 1024>>>(in, out, groups, 0); });
      printf("hints %-22s cfg5 shape %.1f us = %.0f GB/s | prod29 shape %.1f us = %.0f GB/s\n", tag, t5, (double)groups * (1225 + 961) * 16 / t5 / 1e3, t9,
             (double)groups * (841 + 625) * 16 / t9 / 1e3);
    };
    nt(std::integral_constant<int, 0>{}, "none");
    nt(std::integral_constant<int, 1>{}, "nontemporal loads");
    nt(std::integral_constant<int, 2>{}, "nontemporal stores");
    nt(std::integral_constant<int, 3>{}, "both");
  }
  shape<1225, 961>("cfg5 35x35 -> 31x31 (4 pl)", in, out, groups);
  shape<841, 625>("prod 29x29 -> 25x25 (4 pl)", in, out, groups);
  shape<2450, 1922>("cfg5 (8 planes / WG)", in, out, groups / 2);
  return 0;
}
