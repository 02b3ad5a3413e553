// Shared device/host helpers for libhdn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/hdn_hip.h"

#define HDN_WAVE 64
#define HDN_BLOCK 256

namespace hdn {

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int round_up(int a, int b) { return cdiv(a, b) * b; }

// One-time-per-device bookkeeping (hipFuncSetAttribute, device symbol addresses): the library assumes one process per
// GPU but stays correct when a process drives several devices.
// The entry points are reached through ctypes (GIL released), so two host threads can make their first call at the same
// time: the mask is atomic (a duplicated one-time call is harmless, a lost bit is not).  Device ids beyond 63 are never
// marked done and simply repeat the (cheap, idempotent) setup on every call.
struct PerDeviceOnce {
  std::atomic<unsigned long long> mask{0};
  static int device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
  }
  bool done(int d) const { return d >= 0 && d < 64 && ((mask.load(std::memory_order_acquire) >> d) & 1ull); }
  void set(int d) {
    if (d >= 0 && d < 64) mask.fetch_or(1ull << d, std::memory_order_release);
  }
};

// range_check.hip: HDN_OK, or HDN_E_LIMIT when an fp32 input of a two-fp16-piece kernel leaves fp16's range (HDN_CHECK_RANGE=1 /
// hdn_set_check_range; a no-op otherwise and inside stream captures)
int check_fp16_range(const float* x, long long n, hipStream_t stream, int act_domain = 0);   // act_domain 1: x is already x_real * 2^-8

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? HDN_OK : -(1000 + (int)e);
}

// Copy n floats global -> LDS, linear image.  16-byte path when the source is aligned
// (always the case for whole-tensor torch allocations and full plane groups).
__device__ __forceinline__ void copy_g2l(const float* __restrict__ src, float* dst, int n, int tid) {
  if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
    const int n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = tid; i < n4; i += HDN_BLOCK) d4[i] = s4[i];
    for (int i = (n4 << 2) + tid; i < n; i += HDN_BLOCK) dst[i] = src[i];
  } else {
    for (int i = tid; i < n; i += HDN_BLOCK) dst[i] = src[i];
  }
}

// Copy n floats LDS -> global, linear image.
__device__ __forceinline__ void copy_l2g(const float* src, float* __restrict__ dst, int n, int tid) {
  if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
    const int n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = tid; i < n4; i += HDN_BLOCK) d4[i] = s4[i];
    for (int i = (n4 << 2) + tid; i < n; i += HDN_BLOCK) dst[i] = src[i];
  } else {
    for (int i = tid; i < n; i += HDN_BLOCK) dst[i] = src[i];
  }
}

// Streaming hints for data that is touched exactly once (the correlation kernels' planes): nontemporal 16-byte loads / stores.
// tools/experiments/ubench_stream.hip: the 5x5 (x) 35x35 traffic shape 156.8 -> 148.5 us, the 5x5 (x) 29x29 shape 98.9 -> 75-89 us.
// HDN_STREAM_HINT: bit 0 = loads, bit 1 = stores (A/B switch; default both).
#ifndef HDN_STREAM_HINT
#define HDN_STREAM_HINT 3
#endif
typedef float hdn_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#if HDN_STREAM_HINT & 1
  const hdn_f4v v = __builtin_nontemporal_load(reinterpret_cast<const hdn_f4v*>(p));
  return float4{v.x, v.y, v.z, v.w};
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_stream(float4* p, const float4& v) {
#if HDN_STREAM_HINT & 2
  __builtin_nontemporal_store(hdn_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<hdn_f4v*>(p));
#else
  *p = v;
#endif
}

// Same copy for a compile-time float count when the workgroup owns a full plane group: every thread issues ALL of
// its 16-byte loads before the first LDS write, so one HBM round trip covers the whole group (the runtime-count
// loop above gets serialised load -> wait -> write by the compiler).
template <int NFLOATS>
__device__ __forceinline__ void copy_g2l_full(const float* __restrict__ src, float* dst, int tid) {
  static_assert(NFLOATS % 4 == 0, "plane groups are multiples of 4 floats");
  constexpr int N4 = NFLOATS / 4;
  constexpr int ITER = cdiv(N4, HDN_BLOCK);
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  float4 r[ITER];
#pragma unroll
  for (int q = 0; q < ITER; ++q) r[q] = ld_stream(s4 + min(tid + q * HDN_BLOCK, N4 - 1));  // unconditional: stays in registers
#pragma unroll
  for (int q = 0; q < ITER; ++q) {
    const int i = tid + q * HDN_BLOCK;
    if (i < N4) d4[i] = r[q];
  }
}

template <int NFLOATS>
__device__ __forceinline__ void copy_l2g_full(const float* src, float* __restrict__ dst, int tid) {
  static_assert(NFLOATS % 4 == 0, "plane groups are multiples of 4 floats");
  constexpr int N4 = NFLOATS / 4;
  constexpr int ITER = cdiv(N4, HDN_BLOCK);
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
  for (int q = 0; q < ITER; ++q) {
    const int i = tid + q * HDN_BLOCK;
    if (i < N4) st_stream(d4 + i, s4[i]);
  }
}

// Individually rounded fp32 operations.  In this toolchain __fmul_rn/__fadd_rn/... are plain operators compiled under
// the default -ffp-contract=fast, so `__fadd_rn(__fmul_rn(a, b), c)` may still become one fma.  These helpers are
// compiled with contraction off: a product feeding a sum stays two roundings, as in the reference's CPU kernels.
#pragma clang fp contract(off)
__device__ __forceinline__ float rn_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float rn_add(float a, float b) { return a + b; }
__device__ __forceinline__ float rn_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float rn_div(float a, float b) { return a / b; }  // IEEE-correct (v_div_scale/fmas/fixup)
#pragma clang fp contract(fast)

__host__ __device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

typedef float float2v __attribute__((ext_vector_type(2)));

// LDS byte address of a __shared__ pointer (low 32 bits of the flat address are the LDS offset).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)); }

// One ds_read2_b32: the pair (lds[addr + 4*O0], lds[addr + 4*O1]) lands in an aligned VGPR pair, ready to be a
// v_pk_fma_f32 operand.  Issued through inline asm so that LLVM cannot merge the two halves with other loads of
// the same dwords and rebuild the pair with v_mov (which it does for the C++ form).  The load is NOT tracked by
// the compiler's s_waitcnt insertion: consume only after lds_wait_all().
template <int O0, int O1>
__device__ __forceinline__ float2v lds_read_pair(uint32_t addr) {
  static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2_b32 offsets are 8-bit dword counts");
  float2v r;
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(O0), "n"(O1) : "memory");
  return r;
}

__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Make `v` opaque at this point so no consumer of it can be scheduled above (cdna guide §5.7 item 3).
__device__ __forceinline__ void pin(float2v& v) { asm volatile("" : "+v"(v)); }

}  // namespace hdn
