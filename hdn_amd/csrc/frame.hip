// Device-resident frame handling of the tracker: the uploaded uint8 BGR frame stays in HBM and the
// crops the networks consume are produced from it by kernels, instead of numpy / OpenCV on the host with an upload per crop.
//
//   hdn_subwindow_f32               SiameseTracker.get_subwindow / get_subwindow_for_homo, hdn/tracker/base_tracker.py:61-213
//                                   (+ optionally get_search_info's normalisation, get_img_info.py:42-70, fused)
//   hdn_frame_warp_perspective_u8   cv2.warpPerspective(img, inv(H_total), BORDER_REPLICATE), hdn_tracker_proj_e2e.py:154
//   hdn_frame_warp_affine_cubic_u8  img_rot_around_center -> cv2.warpAffine(flags=2), hdn/utils/transform.py:69-100
//
// The crop-position arithmetic, the uint8 mean padding and the normalisation are the reference's own numpy arithmetic and
// are pinned to fixtures produced by the reference (tests/golden/frame.npz).  Everything that is OpenCV in the reference
// (cv2.resize inside get_subwindow, the two warps) restates OpenCV 4.x's published 8-bit fixed-point algorithms and is
// PARITY-UNPINNED: OpenCV is in neither the reference tree nor this image.  oracle/frame_oracle.py carries the same
// restatements in numpy; the GPU tests hold these kernels to it bit for bit.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include "hdn_common.h"

#pragma clang fp contract(off)

namespace hdn {

constexpr int FR_MAXC = 4;

// ---- get_subwindow -------------------------------------------------------------------------------------------------
struct ResizeAxis {  // cv::resize, INTER_LINEAR, 8U: source index and 11-bit weights of one destination coordinate
  int s;
  int w0, w1;
};
__device__ __forceinline__ ResizeAxis resize_axis(int d, int dn, int sn) {
  const double scale = 1.0 / ((double)dn / (double)sn);
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f = rn_sub(f, (float)s);
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= sn - 1) { f = 0.f; s = sn - 1; }
  ResizeAxis r;
  r.s = s;
  r.w0 = (int)rintf(rn_mul(rn_sub(1.0f, f), 2048.0f));
  r.w1 = (int)rintf(rn_mul(f, 2048.0f));
  return r;
}

// params (device, float64): [cx, cy, original_sz, avg_0 .. avg_{C-1}]
// blockIdx.y = the frame of a batch (one sequence each, hdn_subwindow_batch_f32): frames contiguous [B,H,W,C], the parameter records
// `params_stride` doubles apart (they may be columns of a wider per-sequence record), the crops contiguous [B, C | 1, model_sz, model_sz].
__global__ __launch_bounds__(HDN_BLOCK) void subwindow_kernel(const uint8_t* __restrict__ frame, const double* __restrict__ params,
                                                              float* __restrict__ out, int H, int W, int C, int model_sz, int mode,
                                                              int params_stride) {
  frame += size_t(blockIdx.y) * H * W * C;
  params += size_t(blockIdx.y) * params_stride;
  out += size_t(blockIdx.y) * (mode ? 1 : C) * model_sz * model_sz;
  const double cx = params[0], cy = params[1], sz = params[2];
  const double c = (sz - 1.0) / 2.0;
  const double xmin_d = floor(cx - c + 0.5), ymin_d = floor(cy - c + 0.5);
  // rows the reference's slice int(ymin'):int(ymax' + 1) yields (fractional sizes: floor(sz), see oracle/frame_oracle.py)
  const double lpad = (double)(long long)fmax(0.0, -xmin_d);
  const int P = (int)(long long)(xmin_d + sz - 1.0 + lpad + 1.0) - (int)(long long)(xmin_d + lpad);
  const int xmin = (int)xmin_d, ymin = (int)ymin_d;
  int fill[FR_MAXC];
#pragma unroll
  for (int q = 0; q < FR_MAXC; ++q) fill[q] = q < C ? (int)(uint8_t)(long long)params[3 + q] : 0;  // uint8(avg): truncation

  auto patch = [&](int py, int px, int ch) -> int {  // the uint8 patch before the resize
    const int fy = ymin + py, fx = xmin + px;
    return (fy >= 0 && fy < H && fx >= 0 && fx < W) ? (int)frame[(size_t(fy) * W + fx) * C + ch] : fill[ch];
  };
  const double mean_i[3] = {118.93, 113.97, 102.60}, std_i[3] = {69.85, 68.81, 72.45};

  for (int pix = blockIdx.x * HDN_BLOCK + threadIdx.x; pix < model_sz * model_sz; pix += gridDim.x * HDN_BLOCK) {
    const int dy = pix / model_sz, dx = pix - dy * model_sz;
    int v[FR_MAXC];
    if (P == model_sz) {  // (cv2.resize to the same size is the identity as well)
#pragma unroll
      for (int q = 0; q < FR_MAXC; ++q) v[q] = q < C ? patch(dy, dx, q) : 0;
    } else {
      const ResizeAxis ax = resize_axis(dx, model_sz, P), ay = resize_axis(dy, model_sz, P);
      const int x1 = min(ax.s + 1, P - 1), y1 = min(ay.s + 1, P - 1);
#pragma unroll
      for (int q = 0; q < FR_MAXC; ++q) {
        if (q < C) {
          const int S0 = patch(ay.s, ax.s, q) * ax.w0 + patch(ay.s, x1, q) * ax.w1;
          const int S1 = patch(y1, ax.s, q) * ax.w0 + patch(y1, x1, q) * ax.w1;
          const int r = (((ay.w0 * (S0 >> 4)) >> 16) + ((ay.w1 * (S1 >> 4)) >> 16) + 2) >> 2;
          v[q] = min(max(r, 0), 255);
        } else {
          v[q] = 0;
        }
      }
    }
    if (mode == 0) {
#pragma unroll
      for (int q = 0; q < FR_MAXC; ++q)
        if (q < C) out[size_t(q) * model_sz * model_sz + pix] = (float)v[q];
    } else {  // get_search_info: mean over the 3 channels of (x - mean) / std, float64, cast to float32 by the caller's .float()
      const double g0 = ((double)v[0] - mean_i[0]) / std_i[0], g1 = ((double)v[1] - mean_i[1]) / std_i[1],
                   g2 = ((double)v[2] - mean_i[2]) / std_i[2];
      out[pix] = (float)(((g0 + g1) + g2) / 3.0);
    }
  }
}

// ---- cv2.warpPerspective, 8U, INTER_LINEAR, BORDER_REPLICATE ----------------------------------------------------------
// A pixel of the BGR frame is 3 bytes at a byte-aligned address; neighbouring taps of a row are contiguous.  One byte-aligned 4-byte
// load (global memory takes them) per 4 bytes instead of one byte load per sample: the warps are bound by the number of loads.
typedef uint32_t u32u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ld4(const uint8_t* p) { return *reinterpret_cast<const u32u*>(p); }
__device__ __forceinline__ int byte_of(uint32_t w, int i) { return (int)((w >> (8 * i)) & 0xffu); }
__device__ __forceinline__ void inv3(const double* m, double* o) {
  const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  const double id = d != 0.0 ? 1.0 / d : 0.0;
  o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
__device__ __forceinline__ int cv_round_sat(double v) { return (int)rint(fmax(-2147483648.0, fmin(2147483647.0, v))); }

__global__ __launch_bounds__(HDN_BLOCK) void frame_warp_perspective_kernel(const uint8_t* __restrict__ src, const double* __restrict__ M,
                                                                           uint8_t* __restrict__ dst, int H, int W, int C, int bw,
                                                                           int m_stride) {
  src += size_t(blockIdx.y) * H * W * C;      // (blockIdx.y = the frame of a batch, each with its own matrix: hdn_frame_warp_perspective_batch_u8)
  dst += size_t(blockIdx.y) * H * W * C;
  M += size_t(blockIdx.y) * m_stride;
  double m[9], mi[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) m[q] = M[q];
  inv3(m, mi);
  const size_t n = size_t(H) * W;
  for (size_t base = size_t(blockIdx.x) * HDN_BLOCK; base < n; base += size_t(gridDim.x) * HDN_BLOCK) {
    const bool valid = base + threadIdx.x < n;
    const size_t pix = valid ? base + threadIdx.x : n - 1;
    const int y = (int)(pix / W), x = (int)(pix - size_t(y) * W);
    const int bx = (x / bw) * bw, x1 = x - bx;
    const double X0 = mi[0] * bx + mi[1] * y + mi[2], Y0 = mi[3] * bx + mi[4] * y + mi[5], W0 = mi[6] * bx + mi[7] * y + mi[8];
    double Wd = W0 + mi[6] * x1;
    Wd = Wd != 0.0 ? 32.0 / Wd : 0.0;
    const int X = cv_round_sat((X0 + mi[0] * x1) * Wd), Y = cv_round_sat((Y0 + mi[3] * x1) * Wd);
    const int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
    const int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    const int x0 = min(max(sx, 0), W - 1), x1c = (int)min(max((long long)sx + 1, 0LL), (long long)W - 1);
    const int y0 = min(max(sy, 0), H - 1), y1c = (int)min(max((long long)sy + 1, 0LL), (long long)H - 1);
    const uint8_t* p00 = src + (size_t(y0) * W + x0) * C;
    const uint8_t* p01 = src + (size_t(y0) * W + x1c) * C;
    const uint8_t* p10 = src + (size_t(y1c) * W + x0) * C;
    const uint8_t* p11 = src + (size_t(y1c) * W + x1c) * C;
    int res[FR_MAXC] = {0, 0, 0, 0};
    if (C == 3 && x1c == x0 + 1 && size_t(y1c) * W + x0 + 2 < n) {
      // the two taps of a row are 6 contiguous bytes: two 4-byte loads per row (the last two bytes read belong to the next pixel,
      // which exists); same integer sums as the byte path below
      const uint32_t a0 = ld4(p00), a1 = ld4(p00 + 4), b0 = ld4(p10), b1 = ld4(p10 + 4);
      const int t00[3] = {byte_of(a0, 0), byte_of(a0, 1), byte_of(a0, 2)}, t01[3] = {byte_of(a0, 3), byte_of(a1, 0), byte_of(a1, 1)};
      const int t10[3] = {byte_of(b0, 0), byte_of(b0, 1), byte_of(b0, 2)}, t11[3] = {byte_of(b0, 3), byte_of(b1, 0), byte_of(b1, 1)};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int acc = t00[q] * w00 + t01[q] * w01 + t10[q] * w10 + t11[q] * w11;
        res[q] = min(max((acc + (1 << 14)) >> 15, 0), 255);
      }
    } else {
      for (int q = 0; q < C; ++q) {
        const int acc = p00[q] * w00 + p01[q] * w01 + p10[q] * w10 + p11[q] * w11;
        res[q] = min(max((acc + (1 << 14)) >> 15, 0), 255);
      }
    }
    if (valid)
      for (int q = 0; q < C; ++q) dst[pix * C + q] = (uint8_t)res[q];
  }
}

// ---- cv2.warpAffine, 8U, INTER_CUBIC, BORDER_REPLICATE ---------------------------------------------------------------
__device__ short g_cubic_itab[32 * 32 * 16];

__global__ __launch_bounds__(HDN_BLOCK) void frame_warp_affine_cubic_kernel(const uint8_t* __restrict__ src, const double* __restrict__ M,
                                                                            uint8_t* __restrict__ dst, int H, int W, int C, int m_stride) {
  src += size_t(blockIdx.y) * H * W * C;      // (blockIdx.y = the frame of a batch: hdn_frame_warp_affine_cubic_batch_u8)
  dst += size_t(blockIdx.y) * H * W * C;
  M += size_t(blockIdx.y) * m_stride;
  // invert the 2x3 matrix as cv::warpAffine does
  double m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5];
  double D = m00 * m11 - m01 * m10;
  D = D != 0.0 ? 1.0 / D : 0.0;
  const double A11 = m11 * D, A22 = m00 * D;
  m00 = A11; m01 *= -D; m10 *= -D; m11 = A22;
  const double b1 = -m00 * m02 - m01 * m12, b2 = -m10 * m02 - m11 * m12;
  m02 = b1; m12 = b2;
  const size_t n = size_t(H) * W;
  for (size_t base = size_t(blockIdx.x) * HDN_BLOCK; base < n; base += size_t(gridDim.x) * HDN_BLOCK) {
    const bool valid = base + threadIdx.x < n;
    const size_t pix = valid ? base + threadIdx.x : n - 1;
    const int y = (int)(pix / W), x = (int)(pix - size_t(y) * W);
    const long long adelta = llrint(m00 * x * 1024.0), bdelta = llrint(m10 * x * 1024.0);
    const long long X0 = llrint((m01 * y + m02) * 1024.0) + 16, Y0 = llrint((m11 * y + m12) * 1024.0) + 16;
    const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const long long sx = (X >> 5) - 1, sy = (Y >> 5) - 1;
    const short* tab = g_cubic_itab + (((int)(Y & 31) * 32 + (int)(X & 31)) << 4);
    int acc[FR_MAXC] = {0, 0, 0, 0};
    if (C == 3 && sx >= 0 && sx + 3 <= (long long)W - 1) {
      // interior columns: the four taps of a row are 12 contiguous bytes = three 4-byte loads (the sums are integer: any order)
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) {
        const int yy = (int)min(max(sy + k1, 0LL), (long long)H - 1);
        const uint8_t* p = src + (size_t(yy) * W + (size_t)sx) * 3;
        const uint32_t u0 = ld4(p), u1 = ld4(p + 4), u2 = ld4(p + 8);
        const int w0 = tab[k1 * 4], w1 = tab[k1 * 4 + 1], w2 = tab[k1 * 4 + 2], w3 = tab[k1 * 4 + 3];
        acc[0] += byte_of(u0, 0) * w0 + byte_of(u0, 3) * w1 + byte_of(u1, 2) * w2 + byte_of(u2, 1) * w3;
        acc[1] += byte_of(u0, 1) * w0 + byte_of(u1, 0) * w1 + byte_of(u1, 3) * w2 + byte_of(u2, 2) * w3;
        acc[2] += byte_of(u0, 2) * w0 + byte_of(u1, 1) * w1 + byte_of(u2, 0) * w2 + byte_of(u2, 3) * w3;
      }
    } else {
#pragma unroll
      for (int k1 = 0; k1 < 4; ++k1) {
        const int yy = (int)min(max(sy + k1, 0LL), (long long)H - 1);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
          const int xx = (int)min(max(sx + k2, 0LL), (long long)W - 1);
          const int w = tab[k1 * 4 + k2];
          const uint8_t* p = src + (size_t(yy) * W + xx) * C;
          for (int q = 0; q < C; ++q) acc[q] += p[q] * w;
        }
      }
    }
    int res[FR_MAXC];
#pragma unroll
    for (int q = 0; q < FR_MAXC; ++q) res[q] = min(max((acc[q] + (1 << 14)) >> 15, 0), 255);
    if (valid)
      for (int q = 0; q < C; ++q) dst[pix * C + q] = (uint8_t)res[q];
  }
}

// BicubicTab_i of OpenCV's initInterTab2D (A = -0.75; short(v * 32768) with the 16 taps' sum forced to 32768)
static void build_cubic_itab(short* it) {
  float t[32][4];
  const float A = -0.75f;
  for (int i = 0; i < 32; ++i) {
    const float x = (float)i / 32.0f;
    t[i][0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    t[i][1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    t[i][2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    t[i][3] = 1.f - t[i][0] - t[i][1] - t[i][2];
  }
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      short* q = it + ((i * 32 + j) << 4);
      int isum = 0;
      for (int k1 = 0; k1 < 4; ++k1)
        for (int k2 = 0; k2 < 4; ++k2) {
          const float v = t[i][k1] * t[j][k2];
          long r = lrintf(v * 32768.0f);
          r = r > 32767 ? 32767 : (r < -32768 ? -32768 : r);
          q[k1 * 4 + k2] = (short)r;
          isum += (int)r;
        }
      if (isum != 32768) {
        const int diff = isum - 32768;
        int Mk = 2 * 4 + 2, mk = 2 * 4 + 2;
        for (int k1 = 2; k1 < 4; ++k1)
          for (int k2 = 2; k2 < 4; ++k2) {
            if (q[k1 * 4 + k2] < q[mk]) mk = k1 * 4 + k2;
            else if (q[k1 * 4 + k2] > q[Mk]) Mk = k1 * 4 + k2;
          }
        if (diff < 0) q[Mk] = (short)(q[Mk] - diff);
        else q[mk] = (short)(q[mk] - diff);
      }
    }
}

static int ensure_cubic_tab() {
  static PerDeviceOnce once;
  static std::mutex mu;
  const int d = PerDeviceOnce::device();
  if (once.done(d)) return HDN_OK;
  std::lock_guard<std::mutex> lock(mu);
  if (once.done(d)) return HDN_OK;
  static short host[32 * 32 * 16];
  static bool built = false;
  if (!built) { build_cubic_itab(host); built = true; }
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_cubic_itab), host, sizeof(host));  // one-time, synchronous
  if (e != hipSuccess) return -(1000 + (int)e);
  once.set(d);
  return HDN_OK;
}

// ---- cv2.remap, 8U data, INTER_LINEAR, BORDER_CONSTANT 0, float32 maps (what cv2.logPolar / cv2.linearPolar call) -----------
// One lane per destination pixel: the map values go to 1/32 px exactly as remap's converter does (cvRound(map * 32), the
// integer part saturated to short), the four taps are weighted with the 15-bit table values (32 - a)(32 - b) * 32 (exact for
// 1/32 steps), taps outside the source are the border value 0, dst = (sum + 2^14) >> 15.  The planes are float32 holding
// uint8 values (what get_subwindow returns); the result is uint8-valued float32 again.
__global__ __launch_bounds__(HDN_BLOCK) void remap_linear_kernel(const float* __restrict__ src, const float* __restrict__ mapx,
                                                                 const float* __restrict__ mapy, float* __restrict__ dst, int C, int Hs,
                                                                 int Ws, int Hd, int Wd) {
  const size_t n = size_t(Hd) * Wd;
  for (size_t pix = size_t(blockIdx.x) * HDN_BLOCK + threadIdx.x; pix < n; pix += size_t(gridDim.x) * HDN_BLOCK) {
    const int X = (int)rintf(mapx[pix] * 32.f), Y = (int)rintf(mapy[pix] * 32.f);   // cvRound(float): round half to even
    const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767), ax = X & 31, ay = Y & 31;
    const int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    const bool x0 = sx >= 0 && sx < Ws, x1 = sx + 1 >= 0 && sx + 1 < Ws, y0 = sy >= 0 && sy < Hs, y1 = sy + 1 >= 0 && sy + 1 < Hs;
    for (int q = 0; q < C; ++q) {
      const float* pl = src + size_t(q) * Hs * Ws;
      const int v00 = x0 && y0 ? (int)pl[size_t(sy) * Ws + sx] : 0, v01 = x1 && y0 ? (int)pl[size_t(sy) * Ws + sx + 1] : 0;
      const int v10 = x0 && y1 ? (int)pl[size_t(sy + 1) * Ws + sx] : 0, v11 = x1 && y1 ? (int)pl[size_t(sy + 1) * Ws + sx + 1] : 0;
      const int acc = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
      dst[size_t(q) * n + pix] = (float)min(max((acc + (1 << 14)) >> 15, 0), 255);
    }
  }
}

static int frame_grid(size_t n) {
  const size_t b = (n + HDN_BLOCK - 1) / HDN_BLOCK;
  return (int)(b < 8192 ? b : 8192);
}

}  // namespace hdn

extern "C" {

int hdn_subwindow_batch_f32(const unsigned char* frames, const double* params, int params_stride, float* out, int B, int H, int W, int C,
                            int model_sz, int mode, void* stream) {
  if (!frames || !params || !out) return HDN_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || model_sz <= 0 || (mode != 0 && mode != 1) || (mode == 1 && C != 3)) return HDN_E_SHAPE;
  if (params_stride < 3 + C) return HDN_E_SHAPE;
  if (C > hdn::FR_MAXC || model_sz > 4096 || (long long)H * W > (1LL << 30) || B > 65535) return HDN_E_LIMIT;
  hipLaunchKernelGGL(hdn::subwindow_kernel, dim3(hdn::frame_grid((size_t)model_sz * model_sz), B), dim3(HDN_BLOCK), 0,
                     static_cast<hipStream_t>(stream), frames, params, out, H, W, C, model_sz, mode, params_stride);
  return hdn::launch_status();
}

int hdn_subwindow_f32(const unsigned char* frame, const double* params, float* out, int H, int W, int C, int model_sz, int mode,
                      void* stream) {
  return hdn_subwindow_batch_f32(frame, params, 3 + (C > 0 ? C : 0), out, 1, H, W, C, model_sz, mode, stream);
}

int hdn_frame_warp_perspective_batch_u8(const unsigned char* src, const double* M, int m_stride, unsigned char* dst, int B, int H, int W, int C,
                                        void* stream) {
  if (!src || !M || !dst) return HDN_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || m_stride < 9) return HDN_E_SHAPE;
  if (C > hdn::FR_MAXC || (long long)H * W > (1LL << 30) || B > 65535) return HDN_E_LIMIT;
  {
    const unsigned char* const se = src + (size_t)B * H * W * C;
    const unsigned char* const de = dst + (size_t)B * H * W * C;
    if (src < de && dst < se) return HDN_E_ALIAS;
  }
  int bh = H < 16 ? H : 16;                      // OpenCV's block walk (BLOCK_SZ 32): it fixes the summation order of x
  int bw = 1024 / bh < W ? 1024 / bh : W;
  hipLaunchKernelGGL(hdn::frame_warp_perspective_kernel, dim3(hdn::frame_grid((size_t)H * W), B), dim3(HDN_BLOCK), 0,
                     static_cast<hipStream_t>(stream), src, M, dst, H, W, C, bw, m_stride);
  return hdn::launch_status();
}

int hdn_frame_warp_perspective_u8(const unsigned char* src, const double* M, unsigned char* dst, int H, int W, int C, void* stream) {
  return hdn_frame_warp_perspective_batch_u8(src, M, 9, dst, 1, H, W, C, stream);
}

int hdn_frame_warp_affine_cubic_batch_u8(const unsigned char* src, const double* M, int m_stride, unsigned char* dst, int B, int H, int W, int C,
                                         void* stream) {
  if (!src || !M || !dst) return HDN_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || m_stride < 6) return HDN_E_SHAPE;
  if (C > hdn::FR_MAXC || (long long)H * W > (1LL << 30) || B > 65535) return HDN_E_LIMIT;
  {
    const unsigned char* const se = src + (size_t)B * H * W * C;
    const unsigned char* const de = dst + (size_t)B * H * W * C;
    if (src < de && dst < se) return HDN_E_ALIAS;
  }
  const int rc = hdn::ensure_cubic_tab();
  if (rc != HDN_OK) return rc;
  hipLaunchKernelGGL(hdn::frame_warp_affine_cubic_kernel, dim3(hdn::frame_grid((size_t)H * W), B), dim3(HDN_BLOCK), 0,
                     static_cast<hipStream_t>(stream), src, M, dst, H, W, C, m_stride);
  return hdn::launch_status();
}

int hdn_frame_warp_affine_cubic_u8(const unsigned char* src, const double* M, unsigned char* dst, int H, int W, int C, void* stream) {
  return hdn_frame_warp_affine_cubic_batch_u8(src, M, 6, dst, 1, H, W, C, stream);
}

int hdn_remap_linear_f32(const float* src, const float* mapx, const float* mapy, float* dst, int C, int Hs, int Ws, int Hd, int Wd,
                         void* stream) {
  if (!src || !mapx || !mapy || !dst) return HDN_E_NULL;
  if (C <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return HDN_E_SHAPE;
  // (C counts planes that share the maps: the channels of one crop, or of a whole batch of crops)
  if (C > 4096 || (long long)Hs * Ws > (1LL << 30) || (long long)Hd * Wd > (1LL << 30)) return HDN_E_LIMIT;
  if (src == dst) return HDN_E_ALIAS;
  hipLaunchKernelGGL(hdn::remap_linear_kernel, dim3(hdn::frame_grid((size_t)Hd * Wd)), dim3(HDN_BLOCK), 0,
                     static_cast<hipStream_t>(stream), src, mapx, mapy, dst, C, Hs, Ws, Hd, Wd);
  return hdn::launch_status();
}

}  // extern "C"
