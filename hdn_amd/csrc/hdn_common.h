// Shared device/host helpers for libhdn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hdn_hip.h"

#define HDN_WAVE 64
#define HDN_BLOCK 256

namespace hdn {

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int round_up(int a, int b) { return cdiv(a, b) * b; }

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

}  // namespace hdn
