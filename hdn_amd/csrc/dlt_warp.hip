// 4-point DLT solve + projective bilinear warp for gfx950.
// Reference: homo_estimator/Deep_homography/Oneline_DLTv1/utils.py:7-67 (DLT_solve),
//            :70-254 (transformer), :257-274 (transform).
//
// DLT: the 8x8 system is solved by 8 lanes (lane r owns row r of [A|b], 9 float64 registers)
// with Gauss-Jordan elimination and partial pivoting; pivot search and pivot-row broadcast are
// wave shuffles, so nothing touches memory.  The reference inverts A in fp32 and multiplies;
// for the production corners its result is within 7e-7 of the exact solution, which is what
// this kernel returns (rounded once to fp32).
//
// Warp: one lane per output pixel (4 per thread), taps gathered from L2 (a 127x127 image is
// 64 KB).  The arithmetic follows the reference's op order with un-fused multiplies/adds where
// the reference's CPU kernels do not fuse, so that tap selection and the degenerate cases
// (both taps clamped to one index -> exact cancellation, |t| < 1e-7 nudge, float->int
// conversion of out-of-range coordinates) come out as the reference's PyTorch-CPU path gives
// them.  See DESIGN.md "warp semantics".
#include "hdn_common.h"

// The reference's CPU kernels round after every multiply and add here: rn_mul/rn_add/rn_sub/rn_div (hdn_common.h) are
// compiled with contraction off; fusion is spelled out with __builtin_fmaf where the reference fuses.
#pragma clang fp contract(off)

namespace hdn {

constexpr int WARP_PX_PER_THREAD = 4;
constexpr int WARP_PX_PER_BLOCK = HDN_BLOCK * WARP_PX_PER_THREAD;

// reference point order inside the solve: utils.py:18-26 gathers columns [0,1,2,3,6,7,4,5]
__device__ __forceinline__ int dlt_point(int p) { return p == 2 ? 3 : (p == 3 ? 2 : p); }

// All 64 lanes call this; each aligned group of 8 lanes solves the same system redundantly.
// Returns h[r] for r = lane & 7 (the r-th of the 8 unknowns).
__device__ __forceinline__ double dlt_solve_rows(const float* __restrict__ src, const float* __restrict__ off,
                                                 int lane) {
  const int r = lane & 7;
  const int p = dlt_point(r >> 1);
  const double x = (double)src[2 * p], y = (double)src[2 * p + 1];
  // dst = src + off is an fp32 add in the reference (utils.py:36)
  const double u = (double)rn_add(src[2 * p], off[2 * p]);
  const double v = (double)rn_add(src[2 * p + 1], off[2 * p + 1]);
  double a[9];
  if ((r & 1) == 0) {
    a[0] = x; a[1] = y; a[2] = 1.0; a[3] = 0.0; a[4] = 0.0; a[5] = 0.0; a[6] = -u * x; a[7] = -u * y; a[8] = u;
  } else {
    a[0] = 0.0; a[1] = 0.0; a[2] = 0.0; a[3] = x; a[4] = y; a[5] = 1.0; a[6] = -v * x; a[7] = -v * y; a[8] = v;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // partial pivoting over rows >= k; ties resolve to the lowest row (as LAPACK's idamax does)
    double best = (r >= k) ? fabs(a[k]) : -1.0;
    int bidx = r;
#pragma unroll
    for (int m = 4; m >= 1; m >>= 1) {
      const double ob = __shfl_xor(best, m, 8);
      const int oi = __shfl_xor(bidx, m, 8);
      if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    double prow[9], krow[9];
#pragma unroll
    for (int j = k; j < 9; ++j) {
      prow[j] = __shfl(a[j], bidx, 8);
      krow[j] = __shfl(a[j], k, 8);
    }
    if (r == k) {
#pragma unroll
      for (int j = k; j < 9; ++j) a[j] = prow[j];
    } else if (r == bidx) {
#pragma unroll
      for (int j = k; j < 9; ++j) a[j] = krow[j];
    }
    if (r != k) {
      const double f = a[k] / prow[k];
#pragma unroll
      for (int j = k; j < 9; ++j) a[j] = fma(-f, prow[j], a[j]);
    }
  }
  // after Gauss-Jordan row r is [0 .. a[r] .. 0 | a[8]]
  double diag = a[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) diag = (r == j) ? a[j] : diag;
  return a[8] / diag;
}

// The same elimination with the 8 x 9 system in LDS and one wave sharing the work (element (r, j) on lane r*9 + j, the last 8
// on a second round): a pivot step is a handful of LDS round trips instead of ~40 dependent wave shuffles of doubles.  Same
// operations on the same values in the same order as dlt_solve_rows => bit-identical H.  Called by ONE wave (64 lanes, uniform
// control flow); A: 72 doubles of LDS owned by that wave.  On return lane r < 8 holds h[r].
__device__ __forceinline__ double dlt_solve_lds(const float* __restrict__ src, const float* __restrict__ off, double* A, int lane) {
  if (lane < 8) {
    const int r = lane, p = dlt_point(r >> 1);
    const double x = (double)src[2 * p], y = (double)src[2 * p + 1];
    const double u = (double)rn_add(src[2 * p], off[2 * p]), v = (double)rn_add(src[2 * p + 1], off[2 * p + 1]);
    double a[9];
    if ((r & 1) == 0) {
      a[0] = x; a[1] = y; a[2] = 1.0; a[3] = 0.0; a[4] = 0.0; a[5] = 0.0; a[6] = -u * x; a[7] = -u * y; a[8] = u;
    } else {
      a[0] = 0.0; a[1] = 0.0; a[2] = 0.0; a[3] = x; a[4] = y; a[5] = 1.0; a[6] = -v * x; a[7] = -v * y; a[8] = v;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) A[r * 9 + j] = a[j];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int e1 = lane, e2 = lane + 64;
  const int r1 = e1 / 9, j1 = e1 - r1 * 9;
  const int r2 = e2 < 72 ? e2 / 9 : 7, j2 = e2 < 72 ? e2 - (e2 / 9) * 9 : 8;   // lanes >= 8 redo element (7, 8): same value
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // partial pivoting over rows >= k, ties to the lowest row (every lane, redundantly: broadcast reads)
    double best = -1.0;
    int p = k;
#pragma unroll
    for (int r = k; r < 8; ++r) {
      const double v = fabs(A[r * 9 + k]);
      if (v > best) { best = v; p = r; }
    }
    const double pk = A[p * 9 + k];
    auto upd = [&](int r, int j) -> double {
      const int sr = (r == k) ? p : (r == p ? k : r);  // the row this element held before rows k and p were swapped
      const double arj = A[sr * 9 + j], ark = A[sr * 9 + k], pj = A[p * 9 + j];
      if (r == k) return pj;                      // the pivot row moves to position k unchanged
      if (j < k) return arj;
      const double f = ark / pk;
      return fma(-f, pj, arj);
    };
    const double n1 = upd(r1, j1), n2 = upd(r2, j2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane's reads of this step precede every write of it
    A[r1 * 9 + j1] = n1;
    A[r2 * 9 + j2] = n2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const int r = lane & 7;
  return A[r * 9 + 8] / A[r * 9 + r];
}

// theta = Minv * H * M with M = [[ax,0,ax],[0,ay,ay],[0,0,1]], each 3x3 product accumulated in
// k order with one fused step per term, which is what the reference's fp32 bmm produces.
__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C[i * 3 + j] = __builtin_fmaf(A[i * 3 + 2], B[6 + j], __builtin_fmaf(A[i * 3 + 1], B[3 + j], rn_mul(A[i * 3], B[j])));
}

__device__ __forceinline__ void normalise_H(const float* Hm, float ax, float ay, float* theta) {
  const float Minv[9] = {rn_div(1.f, ax), 0.f, -1.f, 0.f, rn_div(1.f, ay), -1.f, 0.f, 0.f, 1.f};
  const float M[9] = {ax, 0.f, ax, 0.f, ay, ay, 0.f, 0.f, 1.f};
  float P[9];
  mat3_mul(Minv, Hm, P);
  mat3_mul(P, M, theta);
}

// torch.linspace(-1, 1, n)[i] on the CPU: start + step*i below the midpoint, end - step*(n-1-i)
// above, each with a single rounding (verified bit-for-bit in tests/test_host_logic.py).
__device__ __forceinline__ float grid_coord(int i, int n) {
  const float step = rn_div(2.0f, (float)(n - 1));
  return i < n / 2 ? __builtin_fmaf(step, (float)i, -1.0f) : __builtin_fmaf(-step, (float)(n - 1 - i), 1.0f);
}

// float -> int32 as the reference's x86 path converts it: out-of-range and NaN give INT_MIN.
__device__ __forceinline__ int f2i_x86(float f) {
  return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One output pixel (row i, column j) of image `img` (C planes of H x W, plane stride H*W) written to
// out[(i*W + j)*C + c].
// Returns the pixel's term of the reference's `condition` count: |t| > 1e-7 after the nudge (utils.py:241).
__device__ __forceinline__ bool warp_pixel(const float* __restrict__ img, const float* th, float* __restrict__ out,
                                           int i, int j, int C, int H, int W) {
  const float gx = grid_coord(j, W), gy = grid_coord(i, H);
  // T = theta @ [gx, gy, 1]: fma(t1, gy, t0*gx) + t2   (the reference's sgemm order; utils.py:230)
  const float T0 = rn_add(__builtin_fmaf(th[1], gy, rn_mul(th[0], gx)), th[2]);
  const float T1 = rn_add(__builtin_fmaf(th[4], gy, rn_mul(th[3], gx)), th[5]);
  float t = rn_add(__builtin_fmaf(th[7], gy, rn_mul(th[6], gx)), th[8]);
  // utils.py:236-240: t += 1e-6 * (1 - [|t| >= 1e-7])
  t = rn_add(t, rn_mul(1e-6f, (fabsf(t) >= 1e-7f) ? 0.0f : 1.0f));
  const float xs = rn_div(T0, t), ys = rn_div(T1, t);
  // utils.py:128-129: x = (x_s + 1) * W / 2
  const float x = rn_mul(rn_mul(rn_add(xs, 1.0f), (float)W), 0.5f);
  const float y = rn_mul(rn_mul(rn_add(ys, 1.0f), (float)H), 0.5f);
  const int xf = f2i_x86(floorf(x)), yf = f2i_x86(floorf(y));
  const int x0 = clampi(xf, 0, W - 1), x1 = clampi((int)((unsigned)xf + 1u), 0, W - 1);
  const int y0 = clampi(yf, 0, H - 1), y1 = clampi((int)((unsigned)yf + 1u), 0, H - 1);
  // utils.py:181-184: weights from the CLAMPED taps and the UNCLAMPED coordinate
  const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
  const float wa = rn_mul(rn_sub(x1f, x), rn_sub(y1f, y));
  const float wb = rn_mul(rn_sub(x1f, x), rn_sub(y, y0f));
  const float wc = rn_mul(rn_sub(x, x0f), rn_sub(y1f, y));
  const float wd = rn_mul(rn_sub(x, x0f), rn_sub(y, y0f));
  const size_t HW = size_t(H) * W;
  float* o = out + (size_t(i) * W + j) * C;
  for (int c = 0; c < C; ++c) {
    const float* pl = img + c * HW;
    const float Ia = pl[y0 * W + x0], Ib = pl[y1 * W + x0], Ic = pl[y0 * W + x1], Id = pl[y1 * W + x1];
    // utils.py:185: wa*Ia + wb*Ib + wc*Ic + wd*Id, left to right, no fusion
    o[c] = rn_add(rn_add(rn_add(rn_mul(wa, Ia), rn_mul(wb, Ib)), rn_mul(wc, Ic)), rn_mul(wd, Id));
  }
  return fabsf(t) > 1e-7f;
}

__global__ __launch_bounds__(HDN_BLOCK) void dlt_solve_kernel(const float* __restrict__ src,
                                                              const float* __restrict__ off,
                                                              float* __restrict__ H_out, int B) {
  // one 8-lane group per sample
  const int gid = (blockIdx.x * HDN_BLOCK + threadIdx.x) >> 3;
  const int b = min(gid, B - 1);  // keep every lane in the shuffles
  const int lane = threadIdx.x & (HDN_WAVE - 1);
  const double h = dlt_solve_rows(src + size_t(b) * 8, off + size_t(b) * 8, lane);
  if (gid < B) {
    H_out[size_t(b) * 9 + (lane & 7)] = (float)h;
    if ((lane & 7) == 0) H_out[size_t(b) * 9 + 8] = 1.0f;
  }
}

__global__ __launch_bounds__(HDN_BLOCK) void warp_kernel(const float* __restrict__ img,
                                                         const float* __restrict__ theta, float* __restrict__ out,
                                                         unsigned int* __restrict__ count, int C, int H, int W) {
  const int b = blockIdx.y;
  const size_t HW = size_t(H) * W;
  float th[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) th[q] = theta[size_t(b) * 9 + q];
  const float* im = img + size_t(b) * C * HW;
  float* ob = out + size_t(b) * C * HW;
  const int base = blockIdx.x * WARP_PX_PER_BLOCK + threadIdx.x;
  unsigned int ok = 0;
#pragma unroll
  for (int q = 0; q < WARP_PX_PER_THREAD; ++q) {
    const int pix = base + q * HDN_BLOCK;
    if (pix < H * W) ok += warp_pixel(im, th, ob, pix / W, pix - (pix / W) * W, C, H, W) ? 1u : 0u;
  }
  if (count) {  // kernel-uniform branch: one atomic per wave
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ok += __shfl_xor(ok, m, HDN_WAVE);
    if ((threadIdx.x & (HDN_WAVE - 1)) == 0 && ok) atomicAdd(count, ok);
  }
}

__global__ __launch_bounds__(HDN_BLOCK) void dlt_warp_kernel(const float* __restrict__ h4p,
                                                             const float* __restrict__ off,
                                                             const float* __restrict__ img,
                                                             float* __restrict__ H_out, float* __restrict__ warped,
                                                             int H, int W, long long img_batch_stride) {
  __shared__ float sH[9];
  __shared__ double sA[72];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  if (tid < HDN_WAVE) {  // wave 0 solves (wave-uniform branch)
    const double h = dlt_solve_lds(h4p + size_t(b) * 8, off + size_t(b) * 8, sA, tid);
    if (tid < 8) sH[tid] = (float)h;
    if (tid == 8) sH[8] = 1.0f;
  }
  __syncthreads();
  float Hm[9], th[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) Hm[q] = sH[q];
  if (blockIdx.x == 0 && tid < 9) H_out[size_t(b) * 9 + tid] = Hm[tid];
  normalise_H(Hm, rn_mul((float)W, 0.5f), rn_mul((float)H, 0.5f), th);
  const size_t HW = size_t(H) * W;
  const int base = blockIdx.x * WARP_PX_PER_BLOCK + tid;
#pragma unroll
  for (int q = 0; q < WARP_PX_PER_THREAD; ++q) {
    const int pix = base + q * HDN_BLOCK;
    if (pix < H * W) warp_pixel(img + size_t(b) * img_batch_stride, th, warped + size_t(b) * HW, pix / W, pix - (pix / W) * W, 1, H, W);
  }
}

// sum |a-b| * scale, one 1024-thread workgroup, fixed reduction tree (deterministic).  The 127x127 planes of the
// two scores are 16 elements per thread: every load is issued before the first add, so the kernel costs one
// L2 round trip instead of a chain of them.
constexpr int L1_THREADS = 1024, L1_PER_THREAD = 16;
// Workgroup j scores a against (j ? b1 : b0) into out[j]: the two scores of a frame are one launch.
__global__ __launch_bounds__(L1_THREADS) void l1_score_kernel(const float* __restrict__ a, const float* __restrict__ b0,
                                                              const float* __restrict__ b1, float* __restrict__ out, int n,
                                                              float scale, long long plane_stride) {
  __shared__ float part[L1_THREADS / HDN_WAVE];
  // blockIdx.y = the sample of a batch (hdn_l1_score2_batch_f32): the three planes `plane_stride` floats further, its two scores at out[2 y]
  a += (size_t)blockIdx.y * plane_stride;
  out += 2 * blockIdx.y;
  const float* __restrict__ b = (blockIdx.x ? b1 : b0) + (size_t)blockIdx.y * plane_stride;
  float s = 0.f;
  for (int base = 0; base < n; base += L1_THREADS * L1_PER_THREAD) {
    float av[L1_PER_THREAD], bv[L1_PER_THREAD];
#pragma unroll
    for (int q = 0; q < L1_PER_THREAD; ++q) {
      const int i = min(base + q * L1_THREADS + (int)threadIdx.x, n - 1);
      av[q] = a[i];
      bv[q] = b[i];
    }
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < L1_PER_THREAD; ++q)
      p[q & 3] += (base + q * L1_THREADS + (int)threadIdx.x < n) ? fabsf(av[q] - bv[q]) : 0.f;
    s += (p[0] + p[1]) + (p[2] + p[3]);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, HDN_WAVE);
  if ((threadIdx.x & (HDN_WAVE - 1)) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x < HDN_WAVE) {
    float t = threadIdx.x < L1_THREADS / HDN_WAVE ? part[threadIdx.x] : 0.f;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) t += __shfl_xor(t, m, HDN_WAVE);
    if (threadIdx.x == 0) out[blockIdx.x] = t * scale;
  }
}

}  // namespace hdn

extern "C" {

int hdn_dlt_solve_f32(const float* src, const float* off, float* H_out, int B, void* stream) {
  if (!src || !off || !H_out) return HDN_E_NULL;
  if (B <= 0) return HDN_E_SHAPE;
  if (B > (1 << 24)) return HDN_E_LIMIT;
  const int groups_per_block = HDN_BLOCK / 8;
  hipLaunchKernelGGL(hdn::dlt_solve_kernel, dim3(hdn::cdiv(B, groups_per_block)), dim3(HDN_BLOCK), 0,
                     static_cast<hipStream_t>(stream), src, off, H_out, B);
  return hdn::launch_status();
}

int hdn_warp_count_f32(const float* img, const float* theta, float* out, unsigned int* count_or_null, int B, int C, int H,
                       int W, void* stream) {
  if (!img || !theta || !out) return HDN_E_NULL;
  if (B <= 0 || C <= 0 || H <= 1 || W <= 1) return HDN_E_SHAPE;  // linspace(-1,1,1) has no step
  if (B > 65535 || (long long)H * W > (1LL << 30) || (long long)C * H * W > 0x7fffffffLL) return HDN_E_LIMIT;
  if (out == img) return HDN_E_ALIAS;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (count_or_null) {
    hipError_t e = hipMemsetAsync(count_or_null, 0, sizeof(unsigned int), st);
    if (e != hipSuccess) return -(1000 + (int)e);
  }
  dim3 grid(hdn::cdiv(H * W, hdn::WARP_PX_PER_BLOCK), B);
  hipLaunchKernelGGL(hdn::warp_kernel, grid, dim3(HDN_BLOCK), 0, st, img, theta, out, count_or_null, C, H, W);
  return hdn::launch_status();
}

int hdn_warp_f32(const float* img, const float* theta, float* out, int B, int C, int H, int W, void* stream) {
  return hdn_warp_count_f32(img, theta, out, nullptr, B, C, H, W, stream);
}

int hdn_dlt_warp_strided_f32(const float* h4p, const float* off, const float* img, long long img_batch_stride, float* H_out, float* warped, int B,
                             int H, int W, void* stream) {
  if (!h4p || !off || !img || !H_out || !warped) return HDN_E_NULL;
  if (B <= 0 || H <= 1 || W <= 1 || img_batch_stride < (long long)H * W) return HDN_E_SHAPE;
  if (B > 65535 || (long long)H * W > (1LL << 30)) return HDN_E_LIMIT;
  if (warped == img) return HDN_E_ALIAS;
  dim3 grid(hdn::cdiv(H * W, hdn::WARP_PX_PER_BLOCK), B);
  hipLaunchKernelGGL(hdn::dlt_warp_kernel, grid, dim3(HDN_BLOCK), 0, static_cast<hipStream_t>(stream), h4p, off, img,
                     H_out, warped, H, W, img_batch_stride);
  return hdn::launch_status();
}

int hdn_dlt_warp_f32(const float* h4p, const float* off, const float* img, float* H_out, float* warped, int B, int H,
                     int W, void* stream) {
  return hdn_dlt_warp_strided_f32(h4p, off, img, (long long)H * W, H_out, warped, B, H, W, stream);
}

int hdn_l1_score_f32(const float* a, const float* b, float* out, int n, float scale, void* stream) {
  if (!a || !b || !out) return HDN_E_NULL;
  if (n <= 0) return HDN_E_SHAPE;
  hipLaunchKernelGGL(hdn::l1_score_kernel, dim3(1), dim3(hdn::L1_THREADS), 0, static_cast<hipStream_t>(stream), a, b, b, out,
                     n, scale, 0LL);
  return hdn::launch_status();
}

int hdn_l1_score2_f32(const float* a, const float* b0, const float* b1, float* out2, int n, float scale, void* stream) {
  if (!a || !b0 || !b1 || !out2) return HDN_E_NULL;
  if (n <= 0) return HDN_E_SHAPE;
  hipLaunchKernelGGL(hdn::l1_score_kernel, dim3(2), dim3(hdn::L1_THREADS), 0, static_cast<hipStream_t>(stream), a, b0, b1, out2,
                     n, scale, 0LL);
  return hdn::launch_status();
}

int hdn_l1_score2_batch_f32(const float* a, const float* b0, const float* b1, float* out, int n, long long plane_stride, int B, float scale,
                            void* stream) {
  if (!a || !b0 || !b1 || !out) return HDN_E_NULL;
  if (n <= 0 || B <= 0 || plane_stride < n) return HDN_E_SHAPE;
  if (B > 65535) return HDN_E_LIMIT;
  hipLaunchKernelGGL(hdn::l1_score_kernel, dim3(2, B), dim3(hdn::L1_THREADS), 0, static_cast<hipStream_t>(stream), a, b0, b1, out, n, scale,
                     plane_stride);
  return hdn::launch_status();
}

int hdn_abi_version(void) { return HDN_ABI_VERSION; }

}  // extern "C"
