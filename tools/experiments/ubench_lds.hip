// Micro-benchmark (not product code): what a LONE wave per SIMD (4 waves per CU, as in the north FFT kernel) pays for the
// LDS access patterns of a 64x64 complex transposition.  s_memtime around blocks of DS instructions, with and without the
// closing s_waitcnt.  hipcc --offload-arch=gfx950 -O3 -o ubench_lds ubench_lds.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <utility>
typedef float cf __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__

template <int I, int E, class F>
DEV void sfor(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, E>(static_cast<F&&>(f));
  }
}
DEV uint64_t now() {
  uint64_t t;
  asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
DEV void wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }


template <int O0, int O1> DEV void w2x64(uint32_t a, cf v0, cf v1) { asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a), "v"(v0), "v"(v1), "n"(O0), "n"(O1)); }
template <int OFF> DEV void w64(uint32_t a, cf v) { asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF)); }
template <int OFF> DEV void wtid(float v) { asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v), "n"(OFF)); }
template <int OFF> DEV void w128(uint32_t a, f4v v) { asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF)); }
template <int OFF> DEV cf r64(uint32_t a) { cf r; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); return r; }
template <int OFF> DEV float r32(uint32_t a) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); return r; }
template <int OFF> DEV f4v r128(uint32_t a) { f4v r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF)); return r; }
template <int O0, int O1> DEV cf r2x32(uint32_t a) { cf r; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(a), "n"(O0), "n"(O1)); return r; }
template <int O0, int O1> DEV f4v r2x64(uint32_t a) { f4v r; asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(a), "n"(O0), "n"(O1)); return r; }

constexpr int WAVE_LDS = 36864;

// MODE: 0 32x ds_write2_b64, lane stride 520 B (current X transposition writes)
//       1 64x ds_write_b64, lane-contiguous (8 B per lane), row stride 496 B
//       2 128x ds_write_addtid_b32, row stride 260 B
//       3 32x ds_write_b128, lane stride 528 B
//       4 61x ds_read_b64 lane-contiguous, row stride 520 B (current column reads)
//       5 31x ds_read_b128, lane stride 496 B
//       6 61x ds_read2_b32, lane stride 516 B (re / im rows 256 B apart)
//       7 32x ds_read2_b64, lane stride 520 B
//       8 64x ds_write_b64 lane stride 520 B (row-owner writes)
//       9 122x ds_read_b32 lane stride 4 (contiguous), row stride 260
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t base = wave * WAVE_LDS;
  float* mine = reinterpret_cast<float*>(smem + base);
  for (int i = lane; i < WAVE_LDS / 4; i += 64) mine[i] = i;
  __syncthreads();
  cf v[64];
  for (int i = 0; i < 64; ++i) v[i] = cf{(float)(lane + i), (float)(lane - i)};
  uint64_t t_issue = 0, t_total = 0;
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
    const uint64_t t0 = now();
    uint64_t t1;
    if constexpr (MODE == 0) {
      const uint32_t a = base + lane * 520;
      sfor<0, 32>([&](auto I) {
        constexpr int i = decltype(I)::value;
        w2x64<2 * i, 2 * i + 1>(a, v[2 * i], v[2 * i + 1]);
      });
    } else if constexpr (MODE == 1) {
      const uint32_t a = base + lane * 8;
      sfor<0, 64>([&](auto I) {
        constexpr int i = decltype(I)::value;
        w64<i * 496>(a, v[i]);
      });
    } else if constexpr (MODE == 2) {
      asm volatile("s_mov_b32 m0, %0" ::"s"(base) : "memory");
      sfor<0, 64>([&](auto I) {
        constexpr int i = decltype(I)::value;
        wtid<i * 520>(v[i].x);
        wtid<i * 520 + 256>(v[i].y);
      });
    } else if constexpr (MODE == 3) {
      const uint32_t a = base + lane * 528;
      sfor<0, 32>([&](auto I) {
        constexpr int i = decltype(I)::value;
        f4v q = {v[2 * i].x, v[2 * i].y, v[2 * i + 1].x, v[2 * i + 1].y};
        w128<i * 16>(a, q);
      });
    } else if constexpr (MODE == 4) {
      const uint32_t a = base + lane * 8;
      sfor<0, 61>([&](auto I) {
        constexpr int i = decltype(I)::value;
        v[i] = r64<i * 520>(a);
      });
    } else if constexpr (MODE == 5) {
      const uint32_t a = base + lane * 496;
      sfor<0, 31>([&](auto I) {
        constexpr int i = decltype(I)::value;
        f4v q = r128<i * 16>(a);
        v[2 * i] = cf{q.x, q.y};
        v[2 * i + 1] = cf{q.z, q.w};
      });
    } else if constexpr (MODE == 6) {
      const uint32_t a = base + lane * 516;
      sfor<0, 61>([&](auto I) {
        constexpr int i = decltype(I)::value;
        v[i] = r2x32<i, i + 64>(a);
      });
    } else if constexpr (MODE == 7) {
      const uint32_t a = base + lane * 520;
      sfor<0, 32>([&](auto I) {
        constexpr int i = decltype(I)::value;
        f4v q = r2x64<2 * i, 2 * i + 1>(a);
        v[2 * i] = cf{q.x, q.y};
        v[2 * i + 1] = cf{q.z, q.w};
      });
    } else if constexpr (MODE == 8) {
      const uint32_t a = base + lane * 520;
      sfor<0, 64>([&](auto I) {
        constexpr int i = decltype(I)::value;
        w64<i * 8>(a, v[i]);
      });
    } else if constexpr (MODE == 9) {
      const uint32_t a = base + lane * 4;
      sfor<0, 61>([&](auto I) {
        constexpr int i = decltype(I)::value;
        v[i].x = r32<i * 520>(a);
        v[i].y = r32<i * 520 + 256>(a);
      });
    }
    {
      uint64_t t;
      asm volatile("s_memtime %0" : "=s"(t)::"memory");  // issued right behind the last DS op: issue cost only
      wait0();
      t1 = t;
    }
    const uint64_t t2 = now();
    t_issue += t1 - t0;
    t_total += t2 - t0;
    for (int i = 0; i < 64; ++i) sink += v[i].x + v[i].y;
    for (int i = 0; i < 64; ++i) v[i] += cf{1.f, 1.f};
  }
  if (lane == 0) {
    const int w = blockIdx.x * 4 + wave;
    out[3 * w] = (uint32_t)(t_issue / iters);
    out[3 * w + 1] = (uint32_t)(t_total / iters);
    out[3 * w + 2] = __float_as_uint(sink);
  }
}

template <int MODE>
void run(const char* name, uint32_t* out, int nops) {
  const int blocks = 256, iters = 200;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WAVE_LDS);
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 4 * WAVE_LDS, 0, out, iters);
    hipDeviceSynchronize();
  }
  static uint32_t h[3 * 1024];
  hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  double si = 0, st = 0;
  for (int w = 0; w < 1024; ++w) { si += h[3 * w]; st += h[3 * w + 1]; }
  si /= 1024; st /= 1024;
  printf("%-58s issue %7.0f clk (%5.1f / op)   issue + wait %7.0f clk (%5.1f / op)\n", name, si, si / nops, st, st / nops);
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 3 * 1024 * 4);
  run<0>("W  32x ds_write2_b64, lane stride 520 (current)", out, 32);
  run<3>("W  32x ds_write_b128, lane stride 528", out, 32);
  run<8>("W  64x ds_write_b64, lane stride 520", out, 64);
  run<1>("W  64x ds_write_b64, lane-contiguous", out, 64);
  run<2>("W 128x ds_write_addtid_b32", out, 128);
  run<4>("R  61x ds_read_b64, lane-contiguous (current)", out, 61);
  run<5>("R  31x ds_read_b128, lane stride 496", out, 31);
  run<6>("R  61x ds_read2_b32, lane stride 516", out, 61);
  run<7>("R  32x ds_read2_b64, lane stride 520", out, 32);
  run<9>("R 122x ds_read_b32, lane-contiguous", out, 122);
  return 0;
}
