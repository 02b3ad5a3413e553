// Channel-contracting correlation of the (unselected) UPChannelBAN head.
// Reference: xcorr_fast, hdn/core/xcorr.py:26-34 (and xcorr_slow, :10-23, the same arithmetic batch by batch):
//   out[b,o,i,j] = sum_c sum_{u,v} x[b,c,i+u,j+v] * k[b,o*C + c,u,v]      x[B,C,Hx,Wx], k[B,O*C,Hk,Wk] -> out[B,O,Ho,Wo]
// O is 2 (cls) or 4 (loc) in the reference (ban.py:26-29), so the contraction has K = C*Hk*Wk = 6400 but only O output
// columns: far too narrow for an MFMA tile (a 32x32 tile would be 6-12 % used) and at ~18 FLOP/B the operator sits at
// the fp32 ridge anyway; it is not on the production path (0 calls per frame).  One lane per output
// pixel, all O outputs in registers, taps wave-uniform (scalar loads), x taps from L1/L2.
#include "hdn_common.h"

namespace hdn {

constexpr int XF_MAX_O = 8;

template <int O>
__global__ __launch_bounds__(HDN_BLOCK) void xcorr_fast_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                               float* __restrict__ out, int C, int Hx, int Wx, int Hk,
                                                               int Wk) {
  const int b = blockIdx.y;
  const int Ho = Hx - Hk + 1, Wo = Wx - Wk + 1;
  const int pix = blockIdx.x * HDN_BLOCK + threadIdx.x;
  const bool live = pix < Ho * Wo;
  const int p = live ? pix : 0;
  const int i = p / Wo, j = p - i * Wo;
  const float* xb = x + size_t(b) * C * Hx * Wx + i * Wx + j;
  const float* kb = k + size_t(b) * O * C * Hk * Wk;
  float acc[O];
#pragma unroll
  for (int o = 0; o < O; ++o) acc[o] = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* xc = xb + size_t(c) * Hx * Wx;
    const float* kc = kb + size_t(c) * Hk * Wk;
    for (int u = 0; u < Hk; ++u)
      for (int v = 0; v < Wk; ++v) {
        const float xv = xc[u * Wx + v];
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = __builtin_fmaf(xv, kc[size_t(o) * C * Hk * Wk + u * Wk + v], acc[o]);
      }
  }
  if (live) {
#pragma unroll
    for (int o = 0; o < O; ++o) out[(size_t(b) * O + o) * Ho * Wo + pix] = acc[o];
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
  dim3 grid(hdn::cdiv(Ho * Wo, HDN_BLOCK), B);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define HDN_XF_CASE(N)                                                                                       \
  case N:                                                                                                    \
    hipLaunchKernelGGL(hdn::xcorr_fast_kernel<N>, grid, dim3(HDN_BLOCK), 0, s, x, k, out, C, Hx, Wx, Hk, Wk); \
    break;
  switch (O) {
    HDN_XF_CASE(1) HDN_XF_CASE(2) HDN_XF_CASE(3) HDN_XF_CASE(4) HDN_XF_CASE(5) HDN_XF_CASE(6) HDN_XF_CASE(7) HDN_XF_CASE(8)
  }
#undef HDN_XF_CASE
  return hdn::launch_status();
}
