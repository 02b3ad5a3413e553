// How fast does ONE wave issue v_mfma_f32_32x32x16_bf16 when it has 1, 2 or 4 independent accumulator chains, with the accumulators
// in VGPRs (as the compiler allocates them in conv3x3_kernel)?   hipcc --offload-arch=gfx950 -O3 ubench_mfma_chain.hip -o ubench_mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* clk, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
  if (s == 1.2345f) out[0] = s;
}

template <int CHAINS>
void run(int waves_per_simd) {
  const int blocks = 256 * 4 * waves_per_simd, iters = 864 / (6 * CHAINS);
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, blocks * 8);
  for (int r = 0; r < 3; ++r) k<CHAINS><<<blocks, 64>>>(out, clk, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) k<CHAINS><<<blocks, 64>>>(out, clk, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[16]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * 6 * CHAINS;
  printf("chains %d, waves/SIMD %d: %4.0f MFMAs per wave: kernel %.1f us; %.1f shader clk per MFMA per wave; %.0f TFLOP/s\n", CHAINS, waves_per_simd, n,
         ms / 20 * 1e3, h[0] / n, n * blocks * 32768.0 / (ms / 20 * 1e-3) / 1e12);
}
int main() {
  run<1>(1); run<2>(1); run<4>(1); run<1>(2); run<2>(2); run<4>(2);
  return 0;
}
