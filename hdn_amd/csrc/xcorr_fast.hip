// Channel-contracting correlation of the (unselected) UPChannelBAN head.
// Reference: xcorr_fast, hdn/core/xcorr.py:26-34 (and xcorr_slow, :10-23, the same arithmetic batch by batch):
//   out[b,o,i,j] = sum_c sum_{u,v} x[b,c,i+u,j+v] * k[b,o*C + c,u,v]      x[B,C,Hx,Wx], k[B,O*C,Hk,Wk] -> out[B,O,Ho,Wo]
// O is 2 (cls) or 4 (loc) in the reference (ban.py:26-29), so the contraction has K = C*Hk*Wk = 6400 but only O output
// columns: far too narrow for an MFMA tile (a 32x32 tile would be 6-12 % used; the 4x4 f32 forms run at the packed-FMA
// rate) and exact fp32 is required; it is not on the production path (0 calls per frame).
// Work split: a workgroup owns 64 output pixels of one batch element; its 4 waves split the CHANNELS (c = wave mod 4), one
// lane per pixel, all O outputs in registers, the taps of a wave's channel wave-uniform (scalar loads), a channel's whole
// window loaded before its first FMA (TAPS known at compile time for the 5x5 case), the four partial sums added in a fixed
// order through the LDS: B * ceil(Ho*Wo / 64) workgroups (640 for the production shape) instead of B * 3, and 4 independent
// channel streams per pixel instead of one 6,400-long dependent chain.
#include "hdn_common.h"

namespace hdn {

constexpr int XF_MAX_O = 8, XF_WAVES = HDN_BLOCK / HDN_WAVE;

template <int O, int TAPS>  // TAPS = Hk = Wk when known at compile time, 0 = runtime
__global__ __launch_bounds__(HDN_BLOCK) void xcorr_fast_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                               float* __restrict__ out, int C, int Hx, int Wx, int Hk_,
                                                               int Wk_) {
  __shared__ float part[XF_WAVES - 1][O][HDN_WAVE];
  const int Hk = TAPS ? TAPS : Hk_, Wk = TAPS ? TAPS : Wk_;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (HDN_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Ho = Hx - Hk + 1, Wo = Wx - Wk + 1;
  const int pix = blockIdx.x * HDN_WAVE + lane;
  const bool live = pix < Ho * Wo;
  const int p = live ? pix : 0;
  const int i = p / Wo, j = p - i * Wo;
  const float* xb = x + size_t(b) * C * Hx * Wx + i * Wx + j;
  const float* kb = k + size_t(b) * O * C * Hk * Wk;
  float acc[O];
#pragma unroll
  for (int o = 0; o < O; ++o) acc[o] = 0.f;
  for (int c = wave; c < C; c += XF_WAVES) {
    const float* xc = xb + size_t(c) * Hx * Wx;
    const float* kc = kb + size_t(c) * Hk * Wk;  // wave-uniform
    if constexpr (TAPS > 0) {
      float win[TAPS * TAPS];
#pragma unroll
      for (int u = 0; u < TAPS; ++u)
#pragma unroll
        for (int v = 0; v < TAPS; ++v) win[u * TAPS + v] = xc[u * Wx + v];
#pragma unroll
      for (int t = 0; t < TAPS * TAPS; ++t)
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = __builtin_fmaf(win[t], kc[size_t(o) * C * TAPS * TAPS + t], acc[o]);
    } else {
      for (int u = 0; u < Hk; ++u)
        for (int v = 0; v < Wk; ++v) {
          const float xv = xc[u * Wx + v];
#pragma unroll
          for (int o = 0; o < O; ++o) acc[o] = __builtin_fmaf(xv, kc[size_t(o) * C * Hk * Wk + u * Wk + v], acc[o]);
        }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int o = 0; o < O; ++o) part[wave - 1][o][lane] = acc[o];
  }
  __syncthreads();
  if (wave == 0 && live) {
#pragma unroll
    for (int o = 0; o < O; ++o) {
      float s = acc[o];
#pragma unroll
      for (int w = 0; w < XF_WAVES - 1; ++w) s += part[w][o][lane];  // fixed order: deterministic
      out[(size_t(b) * O + o) * Ho * Wo + pix] = s;
    }
  }
}

}  // namespace hdn

extern "C" int hdn_xcorr_fast_f32(const float* x, const float* k, float* out, int B, int C, int O, int Hx, int Wx, int Hk,
                                  int Wk, void* stream) {
  if (!x || !k || !out) return HDN_E_NULL;
  if (B <= 0 || C <= 0 || O <= 0 || Hx <= 0 || Wx <= 0 || Hk <= 0 || Wk <= 0 || Hk > Hx || Wk > Wx) return HDN_E_SHAPE;
  if (O > hdn::XF_MAX_O || B > 65535 || (long long)C * Hx * Wx > 0x7fffffffLL || (long long)O * C * Hk * Wk > 0x7fffffffLL)
    return HDN_E_LIMIT;
  if (out == x || out == k) return HDN_E_ALIAS;
  const int Ho = Hx - Hk + 1, Wo = Wx - Wk + 1;
  dim3 grid(hdn::cdiv(Ho * Wo, HDN_WAVE), B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool five = Hk == 5 && Wk == 5;
#define HDN_XF_CASE(N)                                                                                                  \
  case N:                                                                                                               \
    if (five) hipLaunchKernelGGL((hdn::xcorr_fast_kernel<N, 5>), grid, dim3(HDN_BLOCK), 0, s, x, k, out, C, Hx, Wx, Hk, Wk); \
    else hipLaunchKernelGGL((hdn::xcorr_fast_kernel<N, 0>), grid, dim3(HDN_BLOCK), 0, s, x, k, out, C, Hx, Wx, Hk, Wk);     \
    break;
  switch (O) {
    HDN_XF_CASE(1) HDN_XF_CASE(2) HDN_XF_CASE(3) HDN_XF_CASE(4) HDN_XF_CASE(5) HDN_XF_CASE(6) HDN_XF_CASE(7) HDN_XF_CASE(8)
  }
#undef HDN_XF_CASE
  return hdn::launch_status();
}
