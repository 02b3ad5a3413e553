// What shader clock does the chip sustain under packed-fp32 load?  Every wave runs K independent v_pk_fma_f32 chains for a
// fixed instruction count and records s_memtime (shader clocks) and s_memrealtime (100 MHz) around it.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/ubench_clock.hip -o /tmp/ubench_clock && /tmp/ubench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int K, int LDS_PAD>
__global__ __launch_bounds__(64) void burn(float* out, unsigned long long* ticks, int iters, int nopmix) {
  extern __shared__ float pad[];
  float2v a[K], b = {1.0001f, 0.9999f}, c = {1e-9f, -1e-9f};
#pragma unroll
  for (int i = 0; i < K; ++i) a[i] = float2v{(float)threadIdx.x + i, 1.f};
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < K; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (nopmix) asm volatile("s_nop 0\n\ts_nop 0");
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) s += a[i].x + a[i].y;
  if (threadIdx.x == 0) {
    ticks[4 * blockIdx.x] = t1 - t0;
    ticks[4 * blockIdx.x + 1] = r1 - r0;
    ticks[4 * blockIdx.x + 2] = r0;
    ticks[4 * blockIdx.x + 3] = r1;
  }
  if (s == 12345.678f) out[0] = s + pad[0];
}

template <int K>
void run(int waves_per_simd, int nopmix) {
  const int blocks = 256 * 4 * waves_per_simd;
  const int lds = 160 * 1024 / (4 * waves_per_simd) - 256;
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 4); hipMalloc(&ticks, blocks * 32);
  hipFuncSetAttribute((const void*)burn<K, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int iters = 40000 / K;
  for (int rep = 0; rep < 3; ++rep) burn<K, 0><<<blocks, 64, lds>>>(out, ticks, iters, nopmix);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  burn<K, 0><<<blocks, 64, lds>>>(out, ticks, iters, nopmix);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), ticks, blocks * 32, hipMemcpyDeviceToHost);
  double sc = 0, rt = 0, rmin = 1e30, rmax = 0;
  unsigned long long s0 = ~0ull, s1 = 0, e0w = ~0ull, e1w = 0;
  for (int i = 0; i < blocks; ++i) {
    sc += h[4 * i]; rt += h[4 * i + 1];
    rmin = h[4 * i + 1] < rmin ? h[4 * i + 1] : rmin; rmax = h[4 * i + 1] > rmax ? h[4 * i + 1] : rmax;
    s0 = h[4 * i + 2] < s0 ? h[4 * i + 2] : s0; s1 = h[4 * i + 2] > s1 ? h[4 * i + 2] : s1;
    e0w = h[4 * i + 3] < e0w ? h[4 * i + 3] : e0w; e1w = h[4 * i + 3] > e1w ? h[4 * i + 3] : e1w;
  }
  sc /= blocks; rt /= blocks;
  printf("   wave life min %.1f / max %.1f us; first..last start spread %.1f us; first..last end spread %.1f us; first start -> last end %.1f us\n",
         rmin / 100.0, rmax / 100.0, (s1 - s0) / 100.0, (e1w - e0w) / 100.0, (e1w - s0) / 100.0);
  const double n = (double)iters * 16 * K;
  printf("K=%d waves/SIMD=%d nopmix=%d: kernel %.1f us; per wave %.0f shader clk in %.1f us => %.0f MHz; %.2f clk per pk_fma per wave, %.2f ns per SIMD issue\n",
         K, waves_per_simd, nopmix, ms * 1e3, sc, rt / 100.0, sc / (rt / 100.0), sc / n, (rt / 100.0) * 1e3 / n / waves_per_simd);
  hipFree(out); hipFree(ticks);
}

int main() {
  for (int w : {1, 2, 4}) { run<4>(w, 0); run<8>(w, 0); }
  run<8>(1, 1); run<8>(2, 1);
  return 0;
}
