// Micro-benchmark (not product code): issue cost of DPP wave_shr:1 / row_shr:1, v_perm_b32, and MFMA + filler mixes on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 7 + i;
  f32x16 acc[3] = {};
  u32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 64 independent-ish wave_shr:1
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_update_dpp(0u, v[i], 0x138, 0xf, 0xf, true);
    } else if (MODE == 1) {  // row_shr:1
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_update_dpp(0u, v[i], 0x111, 0xf, 0xf, true);
    } else if (MODE == 2) {  // v_perm
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_perm(v[i], v[(i + 1) & 7], 0x05040100u);
    } else if (MODE == 3) {  // 24 MFMAs alone (3 accumulators)
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
    } else if (MODE == 4 || MODE == 5) {  // 24 MFMAs + 56 dpp + 48 perm interleaved (MODE 5: row_shr instead of wave_shr)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 7; ++i) v[i] = __builtin_amdgcn_update_dpp(0u, v[i], MODE == 4 ? 0x138 : 0x111, 0xf, 0xf, true);
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = __builtin_amdgcn_perm(v[i], v[i + 1], 0x05040100u);
      }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int q = 0; q < 3; ++q) for (int g = 0; g < 16; ++g) s += (unsigned)acc[q][g];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, unsigned* out, int blocks, double ops_per_iter) {
  const int iters = 500;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters); hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 3; ++r) { hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  // cycles per op per SIMD at 2.4 GHz: waves per SIMD = blocks*4/1024
  double waves_per_simd = blocks * 4.0 / 1024.0;
  double cyc = best * 1e-3 * 2.4e9 / (iters * ops_per_iter * waves_per_simd);
  printf("%-44s blocks=%4d  %8.3f ms  ~%.1f clk per op per SIMD (@2.4GHz)\n", name, blocks, best, cyc);
}

int main() {
  unsigned* out; hipMalloc(&out, 2048 * 256 * 4);
  for (int blocks : {256, 512, 1024}) {
    run<0>("dpp wave_shr:1 x64", out, blocks, 64);
    run<1>("dpp row_shr:1 x64", out, blocks, 64);
    run<2>("v_perm_b32 x64", out, blocks, 64);
    run<3>("mfma 32x32x16 bf16 x24 (3 acc)", out, blocks, 24);
    run<4>("mfma x24 + 56 wave_shr + 48 perm (per mfma)", out, blocks, 24);
    run<5>("mfma x24 + 56 row_shr + 48 perm (per mfma)", out, blocks, 24);
  }
  return 0;
}
