// Log-polar resample of the search crop.
// Reference: STN_Polar.forward, hdn/models/logpolar.py:100-124 = grid construction on the CPU every call
// (linspace/exp/cos/sin/meshgrid, then an upload) + F.grid_sample(bilinear, padding_mode='border',
// align_corners=False).  Here the three 1-D factors rho[b], cos(theta[a]), sin(theta[a]) are device-resident
// tables (built once per rotation offset by the host with the reference's own ops, so the transcendental values are
// bit-identical) and one kernel forms the grid point, samples all channels and optionally writes the grid.
// Arithmetic follows PyTorch's CPU grid_sampler step by step (un-fused multiplies/adds):
//   g = (rho*cos + polar) / (size//2);  p = (g + 1) * (size/2) - 0.5;  p = min(size-1, max(p, 0));
//   w = p - floor(p), e = 1 - w (same for rows: n, s);  nw = s*e, ne = s*w, sw = n*e, se = n*w;
//   east / south taps beyond the last index are masked;  out = nw*I_nw + ne*I_ne + sw*I_sw + se*I_se.
#include "hdn_common.h"

// The reference's CPU kernels round after every multiply and add here: rn_mul/rn_add/rn_sub/rn_div (hdn_common.h) are
// compiled with contraction off; fusion is spelled out with __builtin_fmaf where the reference fuses.
#pragma clang fp contract(off)

namespace hdn {

__global__ __launch_bounds__(HDN_BLOCK) void logpolar_kernel(const float* __restrict__ img,
                                                             const float* __restrict__ polar,
                                                             const float* __restrict__ rho,
                                                             const float* __restrict__ cosT,
                                                             const float* __restrict__ sinT, float* __restrict__ out,
                                                             float* __restrict__ grid, int C, int H, int W, int S) {
  const int n = blockIdx.y;
  const int pix = blockIdx.x * HDN_BLOCK + threadIdx.x;
  if (pix >= S * S) return;
  const int a = pix / S, b = pix - a * S;  // a = angle row, b = log-radius column
  const float r = rho[b];
  // logpolar.py:112-116: the x coordinate is divided by sz[2]//2 (= H//2), the y coordinate by sz[3]//2 (= W//2)
  const float gx = rn_div(rn_add(rn_mul(r, cosT[a]), polar[2 * n]), (float)(H / 2));
  const float gy = rn_div(rn_add(rn_mul(r, sinT[a]), polar[2 * n + 1]), (float)(W / 2));
  if (grid) {
    float* g = grid + (size_t(n) * S * S + pix) * 2;
    g[0] = gx;
    g[1] = gy;
  }
  float px = rn_sub(rn_mul(rn_add(gx, 1.0f), rn_mul((float)W, 0.5f)), 0.5f);
  float py = rn_sub(rn_mul(rn_add(gy, 1.0f), rn_mul((float)H, 0.5f)), 0.5f);
  px = fminf((float)(W - 1), fmaxf(0.0f, px));  // NaN -> 0, as the reference's clamp order does
  py = fminf((float)(H - 1), fmaxf(0.0f, py));
  const float xw = floorf(px), yn = floorf(py);
  const float w = rn_sub(px, xw), e = rn_sub(1.0f, w);
  const float nn = rn_sub(py, yn), s = rn_sub(1.0f, nn);
  const float nw = rn_mul(s, e), ne = rn_mul(s, w), sw = rn_mul(nn, e), se = rn_mul(nn, w);
  const int x0 = (int)xw, y0 = (int)yn;
  const bool e_ok = x0 + 1 < W, s_ok = y0 + 1 < H;
  const int x1 = e_ok ? x0 + 1 : x0, y1 = s_ok ? y0 + 1 : y0;
  const size_t HW = size_t(H) * W;
  const float* im = img + size_t(n) * C * HW;
  float* o = out + size_t(n) * C * S * S + pix;
  for (int c = 0; c < C; ++c) {
    const float* pl = im + c * HW;
    const float v_nw = pl[y0 * W + x0];
    const float v_ne = e_ok ? pl[y0 * W + x1] : 0.f;
    const float v_sw = s_ok ? pl[y1 * W + x0] : 0.f;
    const float v_se = (e_ok && s_ok) ? pl[y1 * W + x1] : 0.f;
    o[size_t(c) * S * S] =
        rn_add(rn_add(rn_add(rn_mul(v_nw, nw), rn_mul(v_ne, ne)), rn_mul(v_sw, sw)), rn_mul(v_se, se));
  }
}

}  // namespace hdn

extern "C" int hdn_logpolar_sample_f32(const float* img, const float* polar, const float* rho, const float* cos_theta,
                                       const float* sin_theta, float* out, float* grid_or_null, int B, int C, int H,
                                       int W, int S, void* stream) {
  if (!img || !polar || !rho || !cos_theta || !sin_theta || !out) return HDN_E_NULL;
  if (B <= 0 || C <= 0 || H <= 1 || W <= 1 || S <= 0) return HDN_E_SHAPE;
  if (B > 65535 || (long long)H * W > (1LL << 30) || (long long)S * S > (1LL << 30)) return HDN_E_LIMIT;
  if (out == img) return HDN_E_ALIAS;
  dim3 grid(hdn::cdiv(S * S, HDN_BLOCK), B);
  hipLaunchKernelGGL(hdn::logpolar_kernel, grid, dim3(HDN_BLOCK), 0, static_cast<hipStream_t>(stream), img, polar, rho,
                     cos_theta, sin_theta, out, grid_or_null, C, H, W, S);
  return hdn::launch_status();
}
