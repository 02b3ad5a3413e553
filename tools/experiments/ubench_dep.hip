// Issue rate of (packed) fp32 ops for ONE or TWO waves per SIMD, K independent chains interleaved.
// hipcc --offload-arch=gfx950 -O3 tools/experiments/ubench_dep.hip -o tools/experiments/ubench_dep && tools/experiments/ubench_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
// OP 0: pk_fma vgpr operands; 1: pk_add; 2: pk_fma with SGPR src1; 3: v_fma_f32; 4: pk_fma a,b,c distinct regs (no reuse)
template <int K, int OP>
__global__ __launch_bounds__(64) void chain(float* out, int iters, float c) {
  extern __shared__ float lds[];
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, 1.f};
  const f2 cc = {c, c};
  f2 dd = {c + threadIdx.x, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 96 / K; ++u) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(cc));
        else if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(cc));
        else if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(a[k]) : "s"(cc));
        else if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k].x) : "v"(cc.x));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[k]) : "v"(dd), "v"(cc));
      }
    }
  }
  f2 s = a[0];
  for (int i = 1; i < 8; ++i) s += a[i];
  if (s.x == 12345.f) out[threadIdx.x] = s.y + lds[threadIdx.x];
}
template <int K, int OP>
void run(const char* name, int lds_bytes) {
  float* d; (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 1000, n = (96 / K) * K;
  (void)hipFuncSetAttribute((const void*)chain<K, OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int waves_per_cu = 163840 / lds_bytes;
  const int grid = 256 * waves_per_cu;
  chain<K, OP><<<grid, 64, lds_bytes>>>(d, 10, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  chain<K, OP><<<grid, 64, lds_bytes>>>(d, iters, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / ((double)iters * n);  // ns per instruction per wave
  printf("%-10s waves/SIMD=%d K=%d  %.3f ns/instr/wave  -> %.3f ns per SIMD issue\n", name, waves_per_cu / 4, K, per,
         per / (waves_per_cu / 4));
  (void)hipFree(d);
}
int main() {
  for (int occ = 1; occ <= 4; occ *= 2) {
    const int lds = 163840 / (4 * occ);
    run<1, 0>("pk_fma", lds); run<2, 0>("pk_fma", lds); run<3, 0>("pk_fma", lds); run<4, 0>("pk_fma", lds); run<6, 0>("pk_fma", lds); run<8, 0>("pk_fma", lds);
    run<2, 1>("pk_add", lds); run<4, 1>("pk_add", lds);
    run<2, 2>("pk_fma_s", lds); run<4, 2>("pk_fma_s", lds); run<6, 2>("pk_fma_s", lds);
    run<2, 3>("fma", lds); run<4, 3>("fma", lds);
    run<4, 4>("pk_fma_3r", lds); run<6, 4>("pk_fma_3r", lds);
  }
  return 0;
}
