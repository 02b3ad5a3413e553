// What packed-fp32 rate does the whole chip sustain as a function of VALU duty?  (round 6, the closing measurement of the 31x31 (x) 61x61 item)
//
// The FFT correlation kernel issues 2,764 v_pk_*_f32 per pair of planes in ~20.9 k shader clocks: one wave per SIMD, VALU busy 53 % of the
// time, the rest LDS / scalar / waits.  A second wave per SIMD could issue its packed math in those gaps — IF the clock stayed where it is.
// This micro-benchmark measures exactly that trade on the chip, without the kernel: every wave alternates a burst of independent
// v_pk_fma_f32 (16 x K instructions) with an idle stretch (s_sleep) so that the SIMD's VALU duty is set by the ratio, at 1 or 2 waves per
// SIMD, on all 1,024 SIMDs, for a launch as long as the real one (~100 us, launched back to back) and for a long one (~3 ms).  Per wave it
// records s_memtime (shader clocks) and s_memrealtime (100 MHz): clock = d(shader) / d(real); chip-wide rate = packed instructions / time.
//
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/ubench_pk_duty.hip -o tools/experiments/ubench_pk_duty && tools/experiments/ubench_pk_duty
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float hashf(unsigned x) {      // a float in [1, 2) with 23 pseudo-random mantissa bits
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return __uint_as_float(0x3f800000u | (x >> 9));
}

// LDSN: ds_write_b64 + ds_read_b64 pairs per burst (the kernel: 328 LDS instructions per 2,764 packed ops = 15 per 128);
// RANDOM: operands with random mantissas, multiplier of magnitude ~1 with alternating sign (values stay bounded, every bit toggles).
template <int K, int GAP, int LDSN = 0, bool RANDOM = false>
__global__ __launch_bounds__(64) void burn(float* out, unsigned long long* ticks, int iters, float seed) {
  extern __shared__ float pad[];
  float2v a[K], b = {1.0001f + seed, 0.9999f - seed}, c = {1e-9f, -1e-9f};
#pragma unroll
  for (int i = 0; i < K; ++i) a[i] = float2v{(float)threadIdx.x * 1.37f + i + seed, 1.f - seed * threadIdx.x};
  if constexpr (RANDOM) {
    const unsigned t = threadIdx.x + 64 * blockIdx.x;
    b = float2v{(t & 1) ? -0.99993f : 0.99993f, -0.99991f};                     // |b| < 1: bounded, never decays to zero; the sign flips every op
    c = float2v{hashf(t * 7 + 3) - 1.5f, hashf(t * 7 + 4) - 1.5f};
#pragma unroll
    for (int i = 0; i < K; ++i) a[i] = float2v{hashf(t * 31 + i), -hashf(t * 37 + i)};
  }
  float2v* const img = reinterpret_cast<float2v*>(pad) + threadIdx.x;            // lane-contiguous 8-byte accesses, stride 65 rows as the kernel's image
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < K; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (LDSN > 0) {
        if (u < LDSN) {
          img[(u & 31) * 65] = a[u % K];
          asm volatile("" ::: "memory");
          const float2v r = img[((u + 7) & 31) * 65];
          c.x += r.x * 1e-30f;
        }
      }
    }
    if constexpr (GAP > 0) __builtin_amdgcn_s_sleep(GAP);      // 64 x GAP clocks without issue (the kernel's LDS / wait stretches)
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
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

template <int GAP, int LDSN = 0, bool RANDOM = false>
void run(int waves_per_simd, double target_us) {
  constexpr int K = 8;                              // 128 packed FMAs per burst = 512 clocks of VALU
  const int blocks = 256 * 4 * waves_per_simd;
  const int lds = 160 * 1024 / (4 * waves_per_simd) - 256;      // LDS footprint pins the residency: exactly waves_per_simd per SIMD
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, 4);
  hipMalloc(&ticks, blocks * 32);
  hipFuncSetAttribute((const void*)burn<K, GAP, LDSN, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  // iterations for ~target_us at ~2 GHz: a wave's iteration is max(512 x waves (VALU shared), 512 + 64 GAP) clocks
  const double it_clk = (512.0 * waves_per_simd > 512.0 + 64.0 * GAP) ? 512.0 * waves_per_simd : 512.0 + 64.0 * GAP;
  const int iters = (int)(target_us * 2000.0 / it_clk) + 1;
  const int warm = target_us > 1000 ? 2 : 40;       // back-to-back launches in front: the regime the real launch runs in
  for (int rep = 0; rep < warm; ++rep) burn<K, GAP, LDSN, RANDOM><<<blocks, 64, lds>>>(out, ticks, iters, 1e-3f * rep);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  burn<K, GAP, LDSN, RANDOM><<<blocks, 64, lds>>>(out, ticks, iters, 0.5f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), ticks, blocks * 32, hipMemcpyDeviceToHost);
  double sc = 0, rt = 0;
  unsigned long long s0 = ~0ull, e1w = 0;
  for (int i = 0; i < blocks; ++i) {
    sc += h[4 * i];
    rt += h[4 * i + 1];
    s0 = h[4 * i + 2] < s0 ? h[4 * i + 2] : s0;
    e1w = h[4 * i + 3] > e1w ? h[4 * i + 3] : e1w;
  }
  sc /= blocks;
  rt /= blocks;
  const double n = (double)iters * 16 * K;                        // packed instructions per wave
  const double mhz = sc / (rt / 100.0);
  const double duty = n * 4.0 * waves_per_simd / sc;              // VALU-busy share of the SIMD
  const double span_us = (e1w - s0) / 100.0;
  const double gops = n * blocks / (span_us * 1e3);               // packed wave-instructions per ns, chip-wide
  printf("%s%s waves/SIMD %d  gap %3d  launch %8.1f us  clock %5.0f MHz  VALU duty %5.1f %%  duty x GHz %.3f  chip rate %6.1f G v_pk_fma/s = %5.1f TFLOP/s\n",
         RANDOM ? "random data" : "smooth data", LDSN ? " + 16 LDS write/read pairs per burst" : "", waves_per_simd, GAP, ms * 1e3, mhz, duty * 100.0, duty * mhz * 1e-3, gops, gops * 256e-3);
  hipFree(out);
  hipFree(ticks);
}

int main(int argc, char** argv) {
  const double us = argc > 1 ? atof(argv[1]) : 100.0;
  printf("# launch length ~%.0f us (%s)\n", us, us > 1000 ? "one long launch" : "40 launches back to back in front of the measured one");
  // one wave per SIMD: duty 100 / 67 / 57 / 50 / 40 / 33 / 25 %
  run<0>(1, us);  run<4>(1, us);  run<6>(1, us);  run<8>(1, us);  run<12>(1, us);  run<16>(1, us);  run<24>(1, us);
  // two waves per SIMD, each with the same bursts: the second wave fills the first one's gaps
  run<0>(2, us);  run<4>(2, us);  run<8>(2, us);  run<12>(2, us);  run<16>(2, us);  run<24>(2, us);  run<40>(2, us);
  // what else draws on the clock at the kernel's own operating point (one wave per SIMD, ~50 % duty): operand toggling, LDS traffic
  run<6, 0, true>(1, us);  run<6, 16, false>(1, us);  run<6, 16, true>(1, us);
  run<12, 0, true>(2, us);  run<12, 16, false>(2, us);  run<12, 16, true>(2, us);  run<16, 16, true>(2, us);
  return 0;
}
