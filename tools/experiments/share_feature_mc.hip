// EXPERIMENT (round 5, not built into the library; measured and dropped: profiles/round5_conv3x3.txt section 6).  To try it again: copy to
// hdn_amd/csrc/, add it to HIP_SOURCES in __graft_entry__.py and call launch_sf_mc from hdn_share_feature_f32 for W <= 128.
// PreShareFeature (eval mode) with its middle layer on the matrix cores (many images of width <= 128):
//   3 x (conv3x3 pad 1, no bias -> BatchNorm(running stats) -> ReLU), channels 1 -> 4 -> 8 -> 1.
// Reference: homo_estimator/Deep_homography/Oneline_DLTv1/preprocess/input_feature_extractor.py:3-29.  share_feature.hip has the all-vector forms
// (rows-in-registers kernel: 34 % of the fp32 vector peak, packed-FMA issue; it stays the form for few images and for wider ones).
//
// The 4 -> 8 layer is 72 % of the multiply-adds.  Here it is a GEMM on v_mfma_f32_16x16x32_f16 with fp32 carried as two fp16 pieces (mfma_split.h):
//   row m    = a PAIR of horizontally adjacent layer-2 pixels (x = 2p - 1, 2p),
//   column n = (output channel co, pixel of the pair dx): 8 x 2 = all 16 columns,
//   k        = (ky, input channel ci, kx4): the 4-wide window of layer-1 pixels under the pair, 3 x 4 x 4 = 48 of 64 (two MFMAs deep);
//              B[k][n] = w2[co][ci][ky][kx4 - dx], zero where kx4 - dx is not a tap.
// An A fragment (lane = (pair, k group), 8 consecutive k) is then, for two input channels, FOUR CONSECUTIVE layer-1 pixels starting at an even
// column: two aligned dwords of an fp16 image per channel.  Layers 1 and 3 stay on the vector pipe around it; the three layers of an 8-row tile run
// back to back out of LDS (input 14 rows -> layer 1 12 rows as fp16 pieces -> layer 2 10 rows fp32 -> 8 output rows): 73 KB, two workgroups per CU.
#include <cstdlib>

#include "hdn_common.h"
#include "mfma_split.h"

namespace hdn {
namespace sfm {
using namespace hdn::mc;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int SF_W1 = 0, SF_W2 = 36, SF_W3 = 324, SF_ALPHA = 396, SF_BETA = 409;   // the parameter block (include/hdn_hip.h)
constexpr int R = 8;                                  // output rows per workgroup
constexpr int RIN = R + 6, RA = R + 4, RB = R + 2;    // rows of the input / layer-1 / layer-2 tiles
constexpr int PIN = 136;                              // floats per staged input row: x = -3 .. 130 at index x + 3
constexpr int PA = 80;                                // dwords per layer-1 row (x = -2 .. 129 as halves at index x + 2): = 16 mod 64, ky lands 16 banks on
constexpr int CSA = RA * PA + 16;                     // dwords per layer-1 channel image: two channels on = 32 banks on
constexpr int PB = 132;                               // floats per layer-2 row: x = -1 .. 128 at index x + 1
constexpr int CSB = RB * PB;
constexpr int NPAIR = 65, NTILE = (RB * NPAIR + 15) / 16;   // pixel pairs per layer-2 row; MFMA tiles of 16 pairs per workgroup
constexpr int A_BYTES = 8 * CSA * 4, B_BYTES = 8 * CSB * 4;
constexpr int LDS_BYTES = A_BYTES + B_BYTES;
static_assert(RIN * PIN * 4 <= B_BYTES, "the input tile lives where layer 2 goes later");
static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(HDN_BLOCK, 2) void share_feature_mc_kernel(const float* __restrict__ img, const float* __restrict__ prm, float* __restrict__ out,
                                                                       int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* const sA = reinterpret_cast<unsigned*>(smem);                 // [piece][channel] x CSA dwords (two fp16 each)
  float* const sB = reinterpret_cast<float*>(smem + A_BYTES);             // [channel][row][PB]
  float* const sIn = sB;                                                   // (dead before layer 2 writes)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * R;
  const size_t plane = size_t(blockIdx.y) * H * W;
  const float* __restrict__ src = img + plane;

  // ---- this lane's B fragments of layer 2 (constant for the launch): lane = (column n = 2 co + dx, k group q4); element j = (channel of the pair, kx4)
  const int n16 = lane & 15, q4 = lane >> 4;
  u32x4 bw[2][2];                                                          // [MFMA of the K pair][piece]
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const int ky = mf == 0 ? (q4 & 1) : 2, cp = mf == 0 ? (q4 >> 1) : (q4 & 1);
    const bool live = mf == 0 || q4 < 2;                                   // (the second MFMA carries 16 of its 32 k)
    const int co = n16 >> 1, dx = n16 & 1;
    unsigned h[4], l[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {                                       // elements 2 jj, 2 jj + 1
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * jj + e, cl = j >> 2, kx = (j & 3) - dx, ci = 2 * cp + cl;
        const bool tap = live && kx >= 0 && kx <= 2;
        v[e] = tap ? prm[SF_W2 + (((ci >> 1) * 9 + ky * 3 + (tap ? kx : 0)) * 8 + co) * 2 + (ci & 1)] : 0.f;
      }
      split2x2(v[0], v[1], h[jj], l[jj]);
    }
    bw[mf][0] = u32x4{h[0], h[1], h[2], h[3]};
    bw[mf][1] = u32x4{l[0], l[1], l[2], l[3]};
  }

  // ---- input tile: rows r0 - 3 .. r0 + 10, columns -3 .. 130, zero outside the image
  for (int idx = tid; idx < RIN * 134; idx += HDN_BLOCK) {
    const int r = idx / 134, c = idx - r * 134;
    const int gr = r0 - 3 + r, gc = c - 3;
    float v = 0.f;
    if (gr >= 0 && gr < H && gc >= 0 && gc < W) v = src[gr * W + gc];
    sIn[r * PIN + c] = v;
  }
  __syncthreads();
#if defined(HDN_ABLATION) && defined(SFM_EXP_STOP) && SFM_EXP_STOP == 1
  if (sIn[tid] != 12345.f) return;
#endif

  // ---- layer 1 (1 -> 4) on the vector pipe: groups of 4 pixels, rows r0 - 2 .. r0 + 9, columns -2 .. 129; BN, ReLU, zero outside the image (the next
  //      convolution's padding), split into the two fp16 pieces, 8 bytes per (piece, channel)
  for (int grp = tid; grp < RA * 33; grp += HDN_BLOCK) {
    const int a = grp / 33, g4 = grp - a * 33;
    float in[3][6];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float* p = sIn + (a + ky) * PIN + 4 * g4;
      const f4 v0 = *reinterpret_cast<const f4*>(p);
      const f2 v1 = *reinterpret_cast<const f2*>(p + 4);
      in[ky][0] = v0.x; in[ky][1] = v0.y; in[ky][2] = v0.z; in[ky][3] = v0.w; in[ky][4] = v1.x; in[ky][5] = v1.y;
    }
    const int gr = r0 - 2 + a, x0 = -2 + 4 * g4;
    const bool row_in = gr >= 0 && gr < H;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float w = prm[SF_W1 + (ky * 3 + kx) * 4 + ch];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(in[ky][i + kx], w, acc[i]);
        }
      float y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = fmaxf(__builtin_fmaf(acc[i], prm[SF_ALPHA + ch], prm[SF_BETA + ch]), 0.f);
        y[i] = row_in && x0 + i >= 0 && x0 + i < W ? t : 0.f;
      }
      unsigned h0, l0, h1, l1;
      split2x2(y[0], y[1], h0, l0);
      split2x2(y[2], y[3], h1, l1);
      unsigned* d = sA + ch * CSA + a * PA + 2 * g4;
      *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(d + 4 * CSA) = u32x2{l0, l1};
    }
  }
  __syncthreads();
#if defined(HDN_ABLATION) && defined(SFM_EXP_STOP) && SFM_EXP_STOP == 2
  if (sA[tid] != 12345u) return;
#endif

  // ---- layer 2 (4 -> 8) on the matrix cores: tiles of 16 pixel pairs over the flattened (row, pair) index
  {
    const int co = n16 >> 1, dx = n16 & 1;
    const float al = prm[SF_ALPHA + 4 + co], be = prm[SF_BETA + 4 + co];
    const int ky0 = q4 & 1, cp0 = q4 >> 1, cp1 = q4 & 1;
    for (int T = wave; T < NTILE; T += HDN_BLOCK / 64) {
      const int P = min(T * 16 + n16, RB * NPAIR - 1);                     // this lane's pair as an A row (m = lane & 15)
      const int bq = P / NPAIR, p = P - bq * NPAIR;
      u32x4 ah[2], alo[2];
      {
        const unsigned* a0 = sA + (2 * cp0) * CSA + (bq + ky0) * PA + p;   // MFMA 0: (ky, channel pair) = (q4 & 1, q4 >> 1)
        const unsigned* a1 = sA + (2 * cp1) * CSA + (bq + 2) * PA + p;     // MFMA 1: ky = 2, channel pair q4 & 1 (groups 2, 3 meet zero weights)
        ah[0] = u32x4{a0[0], a0[1], a0[CSA], a0[CSA + 1]};
        alo[0] = u32x4{a0[4 * CSA], a0[4 * CSA + 1], a0[5 * CSA], a0[5 * CSA + 1]};
        ah[1] = u32x4{a1[0], a1[1], a1[CSA], a1[CSA + 1]};
        alo[1] = u32x4{a1[4 * CSA], a1[4 * CSA + 1], a1[5 * CSA], a1[5 * CSA + 1]};
      }
      f32x4 dh = {0.f, 0.f, 0.f, 0.f}, dl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        dl = mfma16(alo[mf], bw[mf][0], dl);
        dh = mfma16(ah[mf], bw[mf][0], dh);
        dl = mfma16(ah[mf], bw[mf][1], dl);
      }
      // D: column = lane & 15 = (co, dx), rows 4 (lane >> 4) + j = pairs T * 16 + 4 q4 + j
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int Pj = T * 16 + 4 * q4 + j;
        const int bj = Pj / NPAIR, pj = Pj - bj * NPAIR;
        const int gr = r0 - 1 + bj, x = 2 * pj - 1 + dx;
        const float v = fmaxf(__builtin_fmaf(dh[j] + dl[j] * LO_UNSCALE, al, be), 0.f);
        if (Pj < RB * NPAIR) sB[co * CSB + bj * PB + x + 1] = gr >= 0 && gr < H && x >= 0 && x < W ? v : 0.f;
      }
    }
  }
  __syncthreads();
#if defined(HDN_ABLATION) && defined(SFM_EXP_STOP) && SFM_EXP_STOP == 3
  if (sB[tid] != 12345.f) return;
#endif

  // ---- layer 3 (8 -> 1) on the vector pipe: 4 pixels per thread, straight to HBM
  {
    const int rr = tid >> 5, x0 = 4 * (tid & 31);
    const int gr = r0 + rr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* p = sB + ci * CSB + (rr + ky) * PB + x0;
        const f4 v0 = *reinterpret_cast<const f4*>(p);
        const f2 v1 = *reinterpret_cast<const f2*>(p + 4);
        const float in[6] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float w = prm[SF_W3 + ((ci >> 1) * 9 + ky * 3 + kx) * 2 + (ci & 1)];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(in[i + kx], w, acc[i]);
        }
      }
    if (gr < H) {
      float* o = out + plane + size_t(gr) * W + x0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (x0 + i < W) o[i] = fmaxf(__builtin_fmaf(acc[i], prm[SF_ALPHA + 12], prm[SF_BETA + 12]), 0.f);
    }
  }
}

}  // namespace sfm

// called by hdn_share_feature_f32 (share_feature.hip) for W <= 128 and enough images to fill the chip
int launch_sf_mc(const float* img, const float* folded, float* out, int B, int H, int W, hipStream_t stream) {
  static PerDeviceOnce attr;
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sfm::share_feature_mc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, sfm::LDS_BYTES);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr.set(dev_);
  }
  hipLaunchKernelGGL(sfm::share_feature_mc_kernel, dim3(cdiv(H, sfm::R), B), dim3(HDN_BLOCK), sfm::LDS_BYTES, stream, img, folded, out, H, W);
  return launch_status();
}

}  // namespace hdn
