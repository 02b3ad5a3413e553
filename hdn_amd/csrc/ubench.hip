// hdn_ubench_copy_f32: the practical HBM ceiling beside the 8 TB/s nominal peak ("builder must also report a measured
// device-copy bandwidth").  A plain 16-byte-per-lane grid-stride copy, nontemporal on both sides, 8,192 workgroups of 256 lanes: the launch
// shape that copied fastest in tools/experiments/ubench_stream.hip (5.8 TB/s read + write at 459 MB; 6.3-6.4 TB/s read only).  bench.py times
// it with HIP events on the launch stream and reports `roofline.measured_copy_GBps` = 2 n 4 bytes / time.
#include "hdn_common.h"

namespace hdn {
namespace {

__global__ __launch_bounds__(HDN_BLOCK) void ubench_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
  const long long stride = (long long)gridDim.x * HDN_BLOCK;
  for (long long i = (long long)blockIdx.x * HDN_BLOCK + threadIdx.x; i < n4; i += stride) st_stream(dst + i, ld_stream(src + i));
}

}  // namespace
}  // namespace hdn

extern "C" int hdn_ubench_copy_f32(const float* src, float* dst, long long n, void* stream) {
  if (!src || !dst) return HDN_E_NULL;
  if (n <= 0 || (n & 3)) return HDN_E_SHAPE;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) return HDN_E_LIMIT;
  const float* const se = src + n;
  const float* const de = dst + n;
  if (src < de && dst < se) return HDN_E_ALIAS;
  const long long n4 = n >> 2;
  const long long want = (n4 + HDN_BLOCK - 1) / HDN_BLOCK;
  const unsigned blocks = (unsigned)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(hdn::ubench_copy_kernel, dim3(blocks), dim3(HDN_BLOCK), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n4);
  return hdn::launch_status();
}
