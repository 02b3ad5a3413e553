// HDN_CHECK_RANGE: the debug guard of the two-fp16-piece kernels (conv3x3.hip, head_conv.hip, head_tail.hip).  They carry an fp32
// value x as fp16(x') + 2^-11 fp16((x' - fp16(x')) 2^11) with x' = x 2^-8 (mfma_split.h), which needs |x| < 65,520 x 256 = 1.67e7: beyond it the first piece is inf and the result
// NaN, where the reference's fp32 convolution (homo_estimator/.../backbone/resnet.py:78-94, hdn/models/head/ban.py:55-66) stays finite.
// The packers check the WEIGHTS on the host; the ACTIVATIONS are device data, so checking them costs a reduction and a host round trip
// per call — off by default (the trunk's BatchNorm-folded, ReLU'd activations are O(10)), on with HDN_CHECK_RANGE=1 or
// hdn_set_check_range(1), and on in the -m gpu test suite.  Inside a stream capture the check is skipped (it synchronises).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "hdn_common.h"

namespace hdn {
namespace {

std::atomic<int> g_check{-1};      // -1: not read from the environment yet

__global__ __launch_bounds__(HDN_BLOCK) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * HDN_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * HDN_BLOCK)
    m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);           // |x| as bits: ordered like the value; NaN sorts above inf
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

}  // namespace

bool check_range_enabled() {
  int v = g_check.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("HDN_CHECK_RANGE");
    v = (e && atoi(e) > 0) ? 1 : 0;
    g_check.store(v, std::memory_order_relaxed);
  }
  return v > 0;
}

// HDN_OK, or HDN_E_LIMIT when some |x[i]| >= 65,520 x 256 (or is NaN).  No-op when the guard is off or the stream is capturing.
int check_fp16_range(const float* x, long long n, hipStream_t stream, int act_domain) {
  if (!check_range_enabled() || !x || n <= 0) return HDN_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return HDN_OK; }
  if (cs != hipStreamCaptureStatusNone) return HDN_OK;
  unsigned* word = nullptr;
  if (hipMallocAsync(reinterpret_cast<void**>(&word), sizeof(unsigned), stream) != hipSuccess) return -(1000 + (int)hipGetLastError());
  hipError_t e = hipMemsetAsync(word, 0, sizeof(unsigned), stream);
  const int blocks = (int)((n + HDN_BLOCK - 1) / HDN_BLOCK < 2048 ? (n + HDN_BLOCK - 1) / HDN_BLOCK : 2048);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(HDN_BLOCK), 0, stream, x, n, word);
  unsigned host = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&host, word, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFreeAsync(word, stream);
  if (e != hipSuccess) return -(1000 + (int)e);
  // 65,520 x 2^8: x 2^-8 is then the first value v_cvt_f16_f32 (round-to-nearest-even) turns into inf; in the scaled domain the stored value is the converted one
  if (host >= (act_domain ? 0x477ff000u : 0x4b7ff000u)) {
    float v;
    memcpy(&v, &host, sizeof v);
    fprintf(stderr, "hdn_amd: HDN_CHECK_RANGE: max |x| = %g over %lld fp32 inputs of a two-fp16-piece kernel (they need |x| < %s): HDN_E_LIMIT\n", (double)v, n,
            act_domain ? "65,520 in the scaled domain" : "16,773,120");
    return HDN_E_LIMIT;
  }
  return HDN_OK;
}

}  // namespace hdn

extern "C" int hdn_set_check_range(int on) {
  const int prev = hdn::check_range_enabled() ? 1 : 0;
  hdn::g_check.store(on ? 1 : 0, std::memory_order_relaxed);
  return prev;
}
