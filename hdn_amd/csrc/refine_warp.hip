// One step of the tracker's homography refinement loop on the device (BASELINE config 5: "2-scale coarse-to-fine").
// Reference: hdn/tracker/hdn_tracker_proj_e2e.py:242-250
//     H_hm = inv(H); H_hm /= H_hm[2,2]                                  (numpy, float32 in / out)
//     search = cv2.warpPerspective(search, inv(H_hm), (127,127), borderMode=cv2.BORDER_REPLICATE)
//     H_hm_comp = H_hm_comp @ H_hm                                      (float64)
// cv2 is a third-party dependency that is absent from the reference tree and from this image (pinned only loosely by the
// reference: INSTALL.md "opencv-python").  The sampler below restates OpenCV 4.x's published INTER_LINEAR
// warpPerspective (modules/imgproc/src/imgwarp.cpp, WarpPerspectiveInvoker + remapBilinear):
//   * the 3x3 matrix is inverted in float64 (it is NOT an inverse map: dst(x,y) = src(M^-1 (x,y,1)));
//   * destination pixels are walked in 64 x 16 blocks; for a pixel at block column bx + x1 of row y
//       X0 = M0*bx + M1*y + M2,  Y0 = M3*bx + M4*y + M5,  W0 = M6*bx + M7*y + M8          (float64, this order)
//       W = W0 + M6*x1;  W = W ? 32/W : 0;  X = cvRound((X0 + M0*x1)*W),  Y = cvRound((Y0 + M3*x1)*W)   (saturated to int32)
//     i.e. source coordinates in 1/32 pixel units, rounded half-to-even;
//   * sx = X >> 5, sy = Y >> 5, fx = (X & 31)/32, fy = (Y & 31)/32; the four weights are float32 products
//     (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx; taps at (sy, sx), (sy, sx+1), (sy+1, sx), (sy+1, sx+1) with every index
//     clamped to the image (BORDER_REPLICATE); the sum runs left to right in the image's own type.
// PARITY UNPINNED: without cv2 there is no reference output to hold this against; oracle/hdn_oracle.py carries the same
// restatement in numpy and the tests compare the two (and check the properties that do not depend on OpenCV: identity,
// integer shifts, border replication, the 1/32-pixel quantisation).  It lives behind its own entry point for that reason.
#include "hdn_common.h"

#pragma clang fp contract(off)

namespace hdn {

__device__ __forceinline__ void inv3_f64(const double* m, double* o) {
  // adjugate / determinant, as cv::invert does for 3x3 (DECOMP_LU special case)
  const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  const double id = d != 0.0 ? 1.0 / d : 0.0;
  o[0] = (m[4] * m[8] - m[5] * m[7]) * id;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = (m[5] * m[6] - m[3] * m[8]) * id;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = (m[3] * m[7] - m[4] * m[6]) * id;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__device__ __forceinline__ int cv_round_sat(double v) {
  v = fmax(-2147483648.0, fmin(2147483647.0, v));
  return (int)rint(v);  // round half to even, as cvRound (lrint) does
}

constexpr int RW_BW = 64, RW_BH = 16;  // OpenCV's block walk for a 127 x 127 INTER_LINEAR warp (BLOCK_SZ 32)

// H[B,9] (fp32, what DLT_solve returned) -> per sample: H_hm = fp32(inv(H) / inv(H)[2,2]); M = fp32(inv(H_hm)) is what
// the reference hands to cv2; cv2 inverts it again in float64.  warped = sampler(search, M); Hcomp <- Hcomp @ H_hm (fp64).
__global__ __launch_bounds__(HDN_BLOCK) void refine_warp_kernel(const float* __restrict__ Hm, const float* __restrict__ img,
                                                                float* __restrict__ out, double* __restrict__ Hcomp,
                                                                int H, int W, int bw, int bh) {
  const int b = blockIdx.y;
  double h[9], t[9], hhm[9], m32[9], mi[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) h[q] = (double)Hm[size_t(b) * 9 + q];
  inv3_f64(h, t);
  const double s = 1.0 / (double)(float)t[8];
#pragma unroll
  for (int q = 0; q < 9; ++q) hhm[q] = (double)(float)((double)(float)t[q] * s);  // float32 values, as numpy holds them
  inv3_f64(hhm, t);
#pragma unroll
  for (int q = 0; q < 9; ++q) m32[q] = (double)(float)t[q];
  inv3_f64(m32, mi);
  if (blockIdx.x == 0 && threadIdx.x == 0 && Hcomp) {
    double c[9], r[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) c[q] = Hcomp[size_t(b) * 9 + q];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) r[i * 3 + j] = c[i * 3] * hhm[j] + c[i * 3 + 1] * hhm[3 + j] + c[i * 3 + 2] * hhm[6 + j];
#pragma unroll
    for (int q = 0; q < 9; ++q) Hcomp[size_t(b) * 9 + q] = r[q];
  }
  const float* im = img + size_t(b) * H * W;
  float* ob = out + size_t(b) * H * W;
  for (int pix = blockIdx.x * HDN_BLOCK + threadIdx.x; pix < H * W; pix += gridDim.x * HDN_BLOCK) {
    const int y = pix / W, x = pix - y * W;
    const int bx = (x / bw) * bw, x1 = x - bx;
    const double X0 = mi[0] * bx + mi[1] * y + mi[2];
    const double Y0 = mi[3] * bx + mi[4] * y + mi[5];
    const double W0 = mi[6] * bx + mi[7] * y + mi[8];
    double Wd = W0 + mi[6] * x1;
    Wd = Wd != 0.0 ? 32.0 / Wd : 0.0;
    const int X = cv_round_sat((X0 + mi[0] * x1) * Wd), Y = cv_round_sat((Y0 + mi[3] * x1) * Wd);
    const int sx = X >> 5, sy = Y >> 5;
    const float fx = (float)(X & 31) * (1.0f / 32.0f), fy = (float)(Y & 31) * (1.0f / 32.0f);
    const float w00 = rn_mul(1.0f - fy, 1.0f - fx), w01 = rn_mul(1.0f - fy, fx), w10 = rn_mul(fy, 1.0f - fx), w11 = rn_mul(fy, fx);
    const int x0 = min(max(sx, 0), W - 1), x1c = min(max((long long)sx + 1, 0LL), (long long)W - 1);
    const int y0 = min(max(sy, 0), H - 1), y1c = min(max((long long)sy + 1, 0LL), (long long)H - 1);
    const float v00 = im[y0 * W + x0], v01 = im[y0 * W + x1c], v10 = im[y1c * W + x0], v11 = im[y1c * W + x1c];
    ob[pix] = rn_add(rn_add(rn_add(rn_mul(v00, w00), rn_mul(v01, w01)), rn_mul(v10, w10)), rn_mul(v11, w11));
  }
}

}  // namespace hdn

extern "C" int hdn_refine_warp_f32(const float* H_mat, const float* search, float* warped, double* H_comp_or_null, int B,
                                   int H, int W, void* stream) {
  if (!H_mat || !search || !warped) return HDN_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0) return HDN_E_SHAPE;
  if (B > 65535 || (long long)H * W > (1LL << 30)) return HDN_E_LIMIT;
  if (warped == search) return HDN_E_ALIAS;
  // OpenCV's block geometry (imgwarp.cpp, WarpPerspectiveInvoker): BLOCK_SZ = 32
  int bh = H < 16 ? H : 16;
  int bw = 1024 / bh < W ? 1024 / bh : W;
  bh = 1024 / bw < H ? 1024 / bw : H;
  const int blocks = hdn::cdiv(H * W, HDN_BLOCK);
  hipLaunchKernelGGL(hdn::refine_warp_kernel, dim3(blocks < 64 ? blocks : 64, B), dim3(HDN_BLOCK), 0,
                     static_cast<hipStream_t>(stream), H_mat, search, warped, H_comp_or_null, H, W, bw, bh);
  return hdn::launch_status();
}
