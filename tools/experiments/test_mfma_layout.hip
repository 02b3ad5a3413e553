// Probe (not product code): confirms on gfx950 hardware the operand layouts the split-bf16 correlation kernel relies on:
//   1. DPP wave_shr:1 with zero fill:      out[l] = in[l-1], out[0] = 0
//   2. v_perm_b32 selectors used to pack bf16 pairs
//   3. v_mfma_f32_32x32x16_bf16:           A lane l holds A[l&31][8*(l>>5)+t], B lane l holds B[8*(l>>5)+t][l&31], t = 0..7;
//                                          D reg g of lane l is D[(g&3) + 8*(g>>2) + 4*(l>>5)][l&31]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned short f2bf(float f) { unsigned u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
__device__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

__global__ void probe(const float* A, const float* B, float* D, unsigned* dpp_out, unsigned* perm_out) {
  const int l = threadIdx.x;
  // 1. DPP
  unsigned v = 100 + l;
  unsigned s = __builtin_amdgcn_update_dpp(0u, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
  dpp_out[l] = s;
  // 2. perm
  unsigned w0 = 0xB1B0A1A0u, w1 = 0xD1D0C1C0u;  // w = (lo16 | hi16<<16)
  if (l == 0) { perm_out[0] = __builtin_amdgcn_perm(w1, w0, 0x05040100u); perm_out[1] = __builtin_amdgcn_perm(w1, w0, 0x07060302u); }
  // 3. MFMA with A (32x16), B (16x32) row-major fp32 inputs converted to bf16
  u32x4 a, b;
  const int i = l & 31, h = l >> 5;
  for (int p = 0; p < 4; ++p) {
    unsigned short a0 = f2bf(A[i * 16 + 8 * h + 2 * p]), a1 = f2bf(A[i * 16 + 8 * h + 2 * p + 1]);
    unsigned short b0 = f2bf(B[(8 * h + 2 * p) * 32 + i]), b1 = f2bf(B[(8 * h + 2 * p + 1) * 32 + i]);
    a[p] = a0 | ((unsigned)a1 << 16);
    b[p] = b0 | ((unsigned)b1 << 16);
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + i] = c[g];
}

int main() {
  std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 2 + (j > 9)) % 13 - 6);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
  float *dA, *dB, *dD; unsigned *dp, *dq;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4); hipMalloc(&dp, 64 * 4); hipMalloc(&dq, 8);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dp, dq);
  unsigned p[64], q[2];
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(p, dp, 256, hipMemcpyDeviceToHost); hipMemcpy(q, dq, 8, hipMemcpyDeviceToHost);
  int bad = 0; for (int l = 0; l < 64; ++l) bad += p[l] != (l == 0 ? 0u : 100u + l - 1);
  printf("dpp wave_shr:1 zero-fill: %s (lane0=%u lane1=%u lane32=%u lane63=%u)\n", bad ? "MISMATCH" : "ok", p[0], p[1], p[32], p[63]);
  printf("perm lo-pack 0x%08x (want c1c0a1a0)  hi-pack 0x%08x (want d1d0b1b0)\n", q[0], q[1]);
  double md = 0; for (int x = 0; x < 1024; ++x) md = fmax(md, fabs(D[x] - R[x]));
  printf("mfma 32x32x16 bf16 layout: max |D - ref| = %g %s\n", md, md == 0 ? "ok" : "MISMATCH");
  return 0;
}
