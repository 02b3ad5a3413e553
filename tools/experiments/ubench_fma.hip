// Micro-benchmark: fp32 FMA issue ceilings on gfx950 (plain v_fma_f32, v_pk_fma_f32, SGPR operand,
// dependency distance).  Not part of the product; used to set the VALU roofline for the 31x31 (x) 61x61 kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/ubench_fma.hip -o /tmp/ubench_fma && /tmp/ubench_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b, int iters) {
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "s"(a), "v"(b));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool SGPR>
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b, int iters) {
  float2v acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = float2v{threadIdx.x * 1e-3f + i, 1.f};
  float2v av = {a, a}, bv = {b, b * 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (SGPR)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(bv), "s"(av));
        else
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(bv), "v"(av));
      }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
double time_ms(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float* out;
  const int blocks_per_cu[] = {1, 2, 4, 8};
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
  const int iters = 2000;
  for (int bpc : blocks_per_cu) {
    const int grid = 256 * bpc;
    const double threads = (double)grid * 256;
#define RUN(NAME, KERN, NACC, FLOP_PER_INSTR)                                                      \
  {                                                                                                \
    double ms = time_ms([&] { hipLaunchKernelGGL(KERN, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 1e-7f, iters); }); \
    double flops = threads * iters * 8.0 * NACC * FLOP_PER_INSTR;                                  \
    printf("%-34s waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s\n", NAME, bpc, ms, flops / ms / 1e9);     \
  }
    RUN("v_fma_f32 (sgpr src) 4 acc", (k_fma<4>), 4, 2)
    RUN("v_fma_f32 (sgpr src) 8 acc", (k_fma<8>), 8, 2)
    RUN("v_fma_f32 (sgpr src) 16 acc", (k_fma<16>), 16, 2)
    RUN("v_pk_fma_f32 vgpr 2 pairs", (k_pkfma<2, false>), 2, 4)
    RUN("v_pk_fma_f32 vgpr 4 pairs", (k_pkfma<4, false>), 4, 4)
    RUN("v_pk_fma_f32 vgpr 8 pairs", (k_pkfma<8, false>), 8, 4)
    RUN("v_pk_fma_f32 sgpr 4 pairs", (k_pkfma<4, true>), 4, 4)
    RUN("v_pk_fma_f32 sgpr 8 pairs", (k_pkfma<8, true>), 8, 4)
    RUN("v_pk_fma_f32 sgpr 16 pairs", (k_pkfma<16, true>), 16, 4)
  }
  return 0;
}
