// Residual-block epilogues of the homography regressor's trunk, fused:
//
//   y = relu(y + bias[c])                 after conv1 of a BasicBlock   (bn1 folded into the conv: bias = folded shift)
//   y = relu(y + bias[c] + residual)      after conv2                   (out += residual; relu)
//
// Replaces, per block of homo_estimator/Deep_homography/Oneline_DLTv1/backbone/resnet.py:78-94 (eval mode, BatchNorm
// folded), the separate bias-add, residual-add and ReLU launches PyTorch issues around the MIOpen convolutions: 83 elementwise
// launches per forward become 32, each ONE pass over the activation (16 bytes per lane, in place).  HBM-bound by construction:
// y read + written once, the residual read once, the per-channel bias from the scalar / L1 path.
// Arithmetic: (y + bias) + residual, individually rounded, then max(., 0) - the order PyTorch's own kernels use
// (conv bias first, `out += residual` second).
#include "hdn_common.h"
#include "mfma_split.h"

#pragma clang fp contract(off)

namespace hdn {

typedef float f4 __attribute__((ext_vector_type(4)));

// NHWC: the channel is the fastest index; C % 4 == 0, so a 16-byte word holds 4 consecutive channels of one pixel.
// NCHW with HW % 4 == 0: a 16-byte word lies inside one (b, c) plane.
template <bool NHWC, bool RES>
__global__ __launch_bounds__(HDN_BLOCK) void bias_act_kernel(f4* __restrict__ y, const f4* __restrict__ res, const float* __restrict__ bias,
                                                             unsigned n4, unsigned C, unsigned HW4) {
  const unsigned stride = gridDim.x * HDN_BLOCK;
  const unsigned C4 = C >> 2, mask = C4 - 1;
  const bool pow2 = (C4 & mask) == 0;  // 64 / 128 / 256 / 512 channels: the channel word is a mask away
  for (unsigned i = blockIdx.x * HDN_BLOCK + threadIdx.x; i < n4; i += stride) {
    f4 v = y[i];
    f4 b;
    if (NHWC) {
      b = reinterpret_cast<const f4*>(bias)[pow2 ? (i & mask) : (i % C4)];
    } else {
      const float s = bias[(i / HW4) % C];
      b = f4{s, s, s, s};
    }
    v = v + b;
    if (RES) v = v + res[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    y[i] = v;
  }
}

// any layout / size: one float per lane
template <bool RES>
__global__ __launch_bounds__(HDN_BLOCK) void bias_act_scalar_kernel(float* __restrict__ y, const float* __restrict__ res,
                                                                    const float* __restrict__ bias, long long n, int C, int HW, int nhwc) {
  const long long stride = (long long)gridDim.x * HDN_BLOCK;
  for (long long i = (long long)blockIdx.x * HDN_BLOCK + threadIdx.x; i < n; i += stride) {
    const int c = nhwc ? (int)(i % C) : (int)((i / HW) % C);
    float v = y[i] + bias[c];
    if (RES) v = v + res[i];
    y[i] = fmaxf(v, 0.f);
  }
}

template <bool NHWC>
static void launch_vec(float* y, const float* res, const float* bias, unsigned n4, unsigned C, unsigned HW4, int blocks, hipStream_t s) {
  f4* y4 = reinterpret_cast<f4*>(y);
  const f4* r4 = reinterpret_cast<const f4*>(res);
  if (res) hipLaunchKernelGGL((bias_act_kernel<NHWC, true>), dim3(blocks), dim3(HDN_BLOCK), 0, s, y4, r4, bias, n4, C, HW4);
  else hipLaunchKernelGGL((bias_act_kernel<NHWC, false>), dim3(blocks), dim3(HDN_BLOCK), 0, s, y4, r4, bias, n4, C, HW4);
}

// AdaptiveAvgPool2d(1) + flatten + Linear(C, O) of the regressor's tail (homo_model_builder.py:161-165) as one launch: one
// workgroup per sample; a thread owns channels tid, tid + 256, ...: the mean over the HW positions (summed in position order, then
// times 1 / HW), its O partial products, then a workgroup reduction per output (wave shuffles, one LDS word per wave and output).
// Replaces at::mean + a hipBLASLt GEMM (5 + 7 us at B = 64, two latency-bound launches at the tracker's B = 1).
constexpr int AF_MAX_OUT = 16, AF_THREADS = 512;
template <bool NHWC>
__global__ __launch_bounds__(AF_THREADS) void avgpool_fc_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                float* __restrict__ out, int C, int HW, int O, float in_unscale) {
  __shared__ float part[AF_THREADS / HDN_WAVE][AF_MAX_OUT];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (HDN_WAVE - 1), wave = tid >> 6;
  const float* xb = x + (size_t)b * C * HW;
  const float inv = (1.0f / (float)HW) * in_unscale;     // (in_unscale = 2^8 for activations of the scaled domain: a power of two, exact)
  float acc[AF_MAX_OUT];
#pragma unroll
  for (int o = 0; o < AF_MAX_OUT; ++o) acc[o] = 0.f;
  // 512 threads: the trunk's 512 channels are ONE pass, and a pass is one round trip (the channel's weights are asked for together with its
  // 16 positions; at B = 1 the launch is a single workgroup and nothing but load latency: it was four round trips, 10 us)
  for (int c = tid; c < C; c += AF_THREADS) {
    float wv[AF_MAX_OUT];
#pragma unroll
    for (int o = 0; o < AF_MAX_OUT; ++o) wv[o] = w[(size_t)min(o, O - 1) * C + c];
    float s = 0.f;
    for (int pb = 0; pb < HW; pb += 16) {     // 16 positions in flight at a time, added in position order
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int p = min(pb + j, HW - 1);
        v[j] = NHWC ? xb[(size_t)p * C + c] : xb[(size_t)c * HW + p];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (pb + j < HW) s += v[j];
    }
    const float m = s * inv;
#pragma unroll
    for (int o = 0; o < AF_MAX_OUT; ++o)
      if (o < O) acc[o] = __builtin_fmaf(m, wv[o], acc[o]);
  }
#pragma unroll
  for (int o = 0; o < AF_MAX_OUT; ++o) {
    float v = acc[o];
#pragma unroll
    for (int d = HDN_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) part[wave][o] = v;
  }
  __syncthreads();
  if (tid < O) {
    float v = bias ? bias[tid] : 0.f;
#pragma unroll
    for (int k = 0; k < AF_THREADS / HDN_WAVE; ++k) v += part[k][tid];
    out[(size_t)b * O + tid] = v;
  }
}

}  // namespace hdn

extern "C" int hdn_avgpool_fc_f32(const float* x, const float* w, const float* bias, float* out, int B, int C, int HW, int O, int nhwc, int in_domain, void* stream) {
  if (!x || !w || !out) return HDN_E_NULL;
  if (B <= 0 || C <= 0 || HW <= 0 || O <= 0) return HDN_E_SHAPE;
  if (O > hdn::AF_MAX_OUT || (long long)B * C * HW > 0x7fffffffLL) return HDN_E_LIMIT;
  if ((const void*)out == (const void*)x) return HDN_E_ALIAS;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nhwc) hipLaunchKernelGGL((hdn::avgpool_fc_kernel<true>), dim3(B), dim3(hdn::AF_THREADS), 0, s, x, w, bias, out, C, HW, O, in_domain ? hdn::mc::ACT_UNSCALE : 1.0f);
  else hipLaunchKernelGGL((hdn::avgpool_fc_kernel<false>), dim3(B), dim3(hdn::AF_THREADS), 0, s, x, w, bias, out, C, HW, O, in_domain ? hdn::mc::ACT_UNSCALE : 1.0f);
  return hdn::launch_status();
}

extern "C" int hdn_bias_relu_f32(float* y, const float* bias, const float* residual, int B, int C, int HW, int nhwc, void* stream) {
  if (!y || !bias) return HDN_E_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return HDN_E_SHAPE;
  if (residual == y) return HDN_E_ALIAS;
  const long long n = (long long)B * C * HW;
  if (n > 0x7fffffffLL) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = hdn::aligned16(y) && (!residual || hdn::aligned16(residual)) && (nhwc ? (C % 4 == 0 && hdn::aligned16(bias)) : (HW % 4 == 0));
  if (vec) {
    const long long n4 = n >> 2, want = (n4 + HDN_BLOCK - 1) / HDN_BLOCK;
    const int blocks = (int)(want < 2048 ? want : 2048);  // 8 workgroups per CU x 256 CUs at most, then a grid-stride loop
    if (nhwc) hdn::launch_vec<true>(y, residual, bias, (unsigned)n4, (unsigned)C, (unsigned)(HW >> 2), blocks, s);
    else hdn::launch_vec<false>(y, residual, bias, (unsigned)n4, (unsigned)C, (unsigned)(HW >> 2), blocks, s);
  } else {
    const long long want = (n + HDN_BLOCK - 1) / HDN_BLOCK;
    const int blocks = (int)(want < 4096 ? want : 4096);
    if (residual) hipLaunchKernelGGL((hdn::bias_act_scalar_kernel<true>), dim3(blocks), dim3(HDN_BLOCK), 0, s, y, residual, bias, n, C, HW, nhwc);
    else hipLaunchKernelGGL((hdn::bias_act_scalar_kernel<false>), dim3(blocks), dim3(HDN_BLOCK), 0, s, y, residual, bias, n, C, HW, nhwc);
  }
  return hdn::launch_status();
}
