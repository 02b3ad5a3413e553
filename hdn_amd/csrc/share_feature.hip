// Fused PreShareFeature (eval mode) for gfx950:
//   3 x (conv3x3 pad 1, no bias -> BatchNorm(running stats) -> ReLU), channels 1 -> 4 -> 8 -> 1.
// Reference: homo_estimator/Deep_homography/Oneline_DLTv1/preprocess/input_feature_extractor.py:3-29.
//
// One workgroup produces a ROWS x COLS output tile; the three layers run back to back out of
// LDS (input tile with a 3-pixel halo, 4-channel and 8-channel intermediates with 2- and
// 1-pixel halos), so HBM sees the image once in and once out (129 KB / image instead of the
// 12x intermediate round trips of three separate conv+BN+ReLU launches).
// Each nn.Conv2d zero-pads ITS OWN input, so intermediate activations that fall outside
// the image are forced to 0 (they are not "what the conv would give on an extended image").
// The 396 weights + 26 BN scale/shift values are wave-uniform: they are read through the
// scalar cache (layout chosen so that each tap's output channels are contiguous).
#include "hdn_common.h"

namespace hdn {

constexpr int SF_ROWS = 4;    // output rows per workgroup
constexpr int SF_COLS = 128;  // output columns per workgroup (127-wide crops: one column tile)

constexpr int SF_W1 = 0, SF_W2 = 36, SF_W3 = 324, SF_ALPHA = 396, SF_BETA = 409;

// LDS tiles (row stride = tile width, columns contiguous so consecutive lanes hit consecutive banks)
constexpr int SF_IN_H = SF_ROWS + 6, SF_IN_W = SF_COLS + 6;
constexpr int SF_A_H = SF_ROWS + 4, SF_A_W = SF_COLS + 4;  // layer-1 output, 4 ch
constexpr int SF_B_H = SF_ROWS + 2, SF_B_W = SF_COLS + 2;  // layer-2 output, 8 ch
constexpr int SF_IN_N = SF_IN_H * SF_IN_W;
constexpr int SF_A_N = SF_A_H * SF_A_W;
constexpr int SF_B_N = SF_B_H * SF_B_W;
constexpr int SF_LDS_FLOATS = SF_IN_N + 4 * SF_A_N + 8 * SF_B_N;

__global__ __launch_bounds__(HDN_BLOCK) void share_feature_kernel(const float* __restrict__ img,
                                                                  const float* __restrict__ prm,
                                                                  float* __restrict__ out, int H, int W) {
  __shared__ float smem[SF_LDS_FLOATS];
  float* s_in = smem;
  float* s_a = smem + SF_IN_N;
  float* s_b = s_a + 4 * SF_A_N;

  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * SF_COLS;  // first output column of the tile
  const int r0 = blockIdx.y * SF_ROWS;  // first output row
  const size_t plane = size_t(blockIdx.z) * H * W;
  const float* __restrict__ src = img + plane;

  // ---- input tile, halo 3, zero outside the image -------------------------------------
  for (int idx = tid; idx < SF_IN_N; idx += HDN_BLOCK) {
    const int r = idx / SF_IN_W, c = idx - r * SF_IN_W;
    const int gr = r0 - 3 + r, gc = c0 - 3 + c;
    float v = 0.f;
    if (gr >= 0 && gr < H && gc >= 0 && gc < W) v = src[gr * W + gc];
    s_in[idx] = v;
  }
  __syncthreads();

  // ---- layer 1: 1 -> 4, halo 2 ---------------------------------------------------------
  for (int idx = tid; idx < SF_A_N; idx += HDN_BLOCK) {
    const int r = idx / SF_A_W, c = idx - r * SF_A_W;
    const int gr = r0 - 2 + r, gc = c0 - 2 + c;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float v = s_in[(r + ky) * SF_IN_W + c + kx];
#pragma unroll
        for (int co = 0; co < 4; ++co) acc[co] = __builtin_fmaf(v, prm[SF_W1 + (ky * 3 + kx) * 4 + co], acc[co]);
      }
    const bool inside = gr >= 0 && gr < H && gc >= 0 && gc < W;
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      const float y = __builtin_fmaf(acc[co], prm[SF_ALPHA + co], prm[SF_BETA + co]);
      s_a[co * SF_A_N + idx] = inside ? fmaxf(y, 0.f) : 0.f;
    }
  }
  __syncthreads();

  // ---- layer 2: 4 -> 8, halo 1 ---------------------------------------------------------
  for (int idx = tid; idx < SF_B_N; idx += HDN_BLOCK) {
    const int r = idx / SF_B_W, c = idx - r * SF_B_W;
    const int gr = r0 - 1 + r, gc = c0 - 1 + c;
    float acc[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) acc[co] = 0.f;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = s_a[ci * SF_A_N + (r + ky) * SF_A_W + c + kx];
#pragma unroll
          for (int co = 0; co < 8; ++co)
            acc[co] = __builtin_fmaf(v, prm[SF_W2 + (ci * 9 + ky * 3 + kx) * 8 + co], acc[co]);
        }
    const bool inside = gr >= 0 && gr < H && gc >= 0 && gc < W;
#pragma unroll
    for (int co = 0; co < 8; ++co) {
      const float y = __builtin_fmaf(acc[co], prm[SF_ALPHA + 4 + co], prm[SF_BETA + 4 + co]);
      s_b[co * SF_B_N + idx] = inside ? fmaxf(y, 0.f) : 0.f;
    }
  }
  __syncthreads();

  // ---- layer 3: 8 -> 1, straight to HBM (columns contiguous across lanes) ----------------
  for (int idx = tid; idx < SF_ROWS * SF_COLS; idx += HDN_BLOCK) {
    const int r = idx / SF_COLS, c = idx - r * SF_COLS;
    const int gr = r0 + r, gc = c0 + c;
    if (gr >= H || gc >= W) continue;
    float acc = 0.f;
#pragma unroll
    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc = __builtin_fmaf(s_b[ci * SF_B_N + (r + ky) * SF_B_W + c + kx], prm[SF_W3 + ci * 9 + ky * 3 + kx], acc);
    const float y = __builtin_fmaf(acc, prm[SF_ALPHA + 12], prm[SF_BETA + 12]);
    out[plane + size_t(gr) * W + gc] = fmaxf(y, 0.f);
  }
}

}  // namespace hdn

extern "C" int hdn_share_feature_f32(const float* img, const float* folded, float* out, int B, int H, int W,
                                     void* stream) {
  if (!img || !folded || !out) return HDN_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0) return HDN_E_SHAPE;
  if (B > 65535 || (long long)H * W > 0x7fffffffLL / 4) return HDN_E_LIMIT;
  if (out == img) return HDN_E_ALIAS;
  dim3 grid(hdn::cdiv(W, hdn::SF_COLS), hdn::cdiv(H, hdn::SF_ROWS), B);
  if (grid.y > 65535) return HDN_E_LIMIT;
  hipLaunchKernelGGL(hdn::share_feature_kernel, grid, dim3(HDN_BLOCK), 0, static_cast<hipStream_t>(stream), img, folded,
                     out, H, W);
  return hdn::launch_status();
}
