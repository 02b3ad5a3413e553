// 3x3 / stride 1 / pad 1 convolution + bias + (residual) + ReLU of the homography regressor's trunk as ONE implicit-GEMM kernel
// on the matrix cores.  Replaces, per BasicBlock of
// homo_estimator/Deep_homography/Oneline_DLTv1/backbone/resnet.py:78-94 (eval mode, BatchNorm folded into the weights), a MIOpen
// fp32 convolution plus the elementwise tail.  29 of the trunk's 36 convolutions have this shape (C -> C channels at S x S,
// (C, S) = (64, 32), (128, 16), (256, 8), (512, 4) for 127-px crops).
//
// GEMM view (NHWC): D[pixel][cout] = sum over (tap, cin) A_tap[pixel][cin] * W[tap][cin][cout], A_tap = the input shifted by the tap.
// fp32 in, fp32 out, but the products run on the bf16 matrix pipe at 16x the fp32 MFMA rate: every fp32 value is split EXACTLY into
// three bf16 pieces x = x0 + x1 + x2 (8 + 8 + 8 significand bits) and six piece products are accumulated in fp32
// (x0 w0, x0 w1, x1 w0, x1 w1, x0 w2, x2 w0; the three dropped ones are below 2^-24 of the product): 6 x 32 clk per K = 16
// against 8 x 64 clk on v_mfma_f32_32x32x2_f32, with a result that differs from an fp32 convolution by summation order and
// < 2^-23 relative per product.  The weights are split once on the host (conv3x3_pack_weights), the activations while they are
// staged into LDS.
//
// Workgroup = 4 waves, tile = BM output pixels (consecutive in (b, y, x) order: whole image rows) x BN output channels.
//   LDS A image: the tile's input pixels with a one-pixel halo (zeros outside the image), one K chunk of 16 input channels at a
//                time, as [piece][k half][pixel] x 16 B: an MFMA A fragment (lane = (pixel row i, k half g), 8 bf16 = 16 B) is
//                one conflict-free ds_read_b128, and a tap is a constant address offset.
//   LDS W image: [piece][k half][cout] x 16 B per tap, double buffered, streamed from the host-packed layout.
//   wave tile  : MT x NT MFMA tiles of 32 x 32; accumulators stay in registers over the whole K loop (C / 16 chunks x 9 taps);
//   epilogue   : + bias[cout] (+ residual) -> ReLU -> NHWC store, 128 contiguous bytes per pixel row and half wave.
#include "hdn_common.h"

namespace hdn {
namespace cv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// round-to-nearest-even fp32 -> bf16 bits (finite inputs)
__device__ __forceinline__ unsigned bf16_rne(float f) {
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// two fp32 values -> their three bf16 pieces, packed (lo = first value)
__device__ __forceinline__ void split3x2(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned a0 = bf16_rne(x), b0 = bf16_rne(y);
  const float xr = x - __uint_as_float(a0 << 16), yr = y - __uint_as_float(b0 << 16);
  const unsigned a1 = bf16_rne(xr), b1 = bf16_rne(yr);
  const float xs = xr - __uint_as_float(a1 << 16), ys = yr - __uint_as_float(b1 << 16);
  const unsigned a2 = bf16_rne(xs), b2 = bf16_rne(ys);
  p0 = a0 | (b0 << 16);
  p1 = a1 | (b1 << 16);
  p2 = a2 | (b2 << 16);
}

template <int S_, int C_, int WM_, int WN_, int MT_, int NT_>
struct Cfg {
  static constexpr int S = S_, C = C_, WM = WM_, WN = WN_, MT = MT_, NT = NT_;
  static constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(C % BN == 0 && C % 16 == 0, "channel blocking");
  static_assert((BM % S == 0) && ((S * S) % BM == 0 || BM % (S * S) == 0), "a tile is whole rows of one image, or whole images");
  static constexpr int IMGS = BM > S * S ? BM / (S * S) : 1;      // images per tile
  static constexpr int R = BM / (S * IMGS);                      // output rows per image in the tile
  static constexpr int PW = S + 2, PH = R + 2;                   // halo'ed image patch
  static constexpr int LP = IMGS * PH * PW;                      // LDS pixels
  static constexpr int KG_BYTES = LP * 16, PIECE_BYTES = 2 * KG_BYTES, A_BYTES = 3 * PIECE_BYTES;
  static constexpr int WKG_BYTES = BN * 16, WPIECE_BYTES = 2 * WKG_BYTES, WTAP_BYTES = 3 * WPIECE_BYTES;
  static constexpr int LDS_BYTES = A_BYTES + 2 * WTAP_BYTES;
  static constexpr int KC = C / 16, NB = C / BN;
  static constexpr int W4 = WTAP_BYTES / 16;                     // 16-byte words of one tap's weights
  static constexpr int WITER = cdiv(W4, HDN_BLOCK);
  static constexpr int AITEMS = LP * 2, AITER = cdiv(AITEMS, HDN_BLOCK);
};

template <class Cf, bool RES>
__global__ __launch_bounds__(HDN_BLOCK) void conv3x3_kernel(const float* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                            const float* __restrict__ res, float* __restrict__ out, int B) {
  constexpr int S = Cf::S, C = Cf::C, MT = Cf::MT, NT = Cf::NT, BM = Cf::BM, BN = Cf::BN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sA = smem;
  unsigned char* const sW = smem + Cf::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cf::WN, wn = wave % Cf::WN;
  const int li = lane & 31, g = lane >> 5;
  const long long m0 = (long long)blockIdx.x * BM;          // first output pixel of the tile, (b, y, x) order
  const int nb = blockIdx.y;                                 // output-channel block
  const int b0 = (int)(m0 / (S * S)), y0 = (int)((m0 % (S * S)) / S);
  const long long M = (long long)B * S * S;

  // ---- per-lane LDS pixel of each of this wave's M tiles (tap (1, 1)); a tap adds a constant
  int lp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int i = (wm * MT + mt) * 32 + li;                  // pixel inside the tile
    const int img = i / (Cf::R * S), yy = (i / S) % Cf::R, xx = i % S;
    lp[mt] = img * (Cf::PH * Cf::PW) + (yy + 1) * Cf::PW + (xx + 1);
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const u32x4* wblock = wp + (size_t)nb * Cf::KC * 9 * Cf::W4;   // this channel block's packed weights: [kc][tap][piece][k half][n][8 bf16]
  auto load_w = [&](u32x4 (&wr)[Cf::WITER], int kc, int tap) {
    const u32x4* src = wblock + (size_t)(kc * 9 + tap) * Cf::W4;
#pragma unroll
    for (int q = 0; q < Cf::WITER; ++q) wr[q] = src[min(tid + q * HDN_BLOCK, Cf::W4 - 1)];
  };
  auto store_w = [&](const u32x4 (&wr)[Cf::WITER], int buf) {
    u32x4* dst = reinterpret_cast<u32x4*>(sW + buf * Cf::WTAP_BYTES);
#pragma unroll
    for (int q = 0; q < Cf::WITER; ++q)
      if (tid + q * HDN_BLOCK < Cf::W4) dst[tid + q * HDN_BLOCK] = wr[q];
  };

  for (int kc = 0; kc < Cf::KC; ++kc) {
    u32x4 wr[Cf::WITER];
    load_w(wr, kc, 0);
    // ---- stage the input chunk: (LDS pixel, k half) items, 8 channels = 32 bytes each, split into the three pieces
    f4 v[Cf::AITER][2];
#pragma unroll
    for (int q = 0; q < Cf::AITER; ++q) {
      const int item = tid + q * HDN_BLOCK;
      const int px = min(item >> 1, Cf::LP - 1), kg = item & 1;
      const int img = px / (Cf::PH * Cf::PW), ry = (px / Cf::PW) % Cf::PH, rx = px % Cf::PW;
      const int b = b0 + img, y = y0 + ry - 1, xx = rx - 1;
      const bool ok = item < Cf::AITEMS && b < B && y >= 0 && y < S && xx >= 0 && xx < S;
      const f4* src = reinterpret_cast<const f4*>(x + (((size_t)b * S + y) * S + xx) * C + kc * 16 + kg * 8);
      v[q][0] = ok ? src[0] : f4{0.f, 0.f, 0.f, 0.f};
      v[q][1] = ok ? src[1] : f4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();  // the previous chunk's fragments have all been read
#pragma unroll
    for (int q = 0; q < Cf::AITER; ++q) {
      const int item = tid + q * HDN_BLOCK;
      if (item < Cf::AITEMS) {
        const int px = item >> 1, kg = item & 1;
        unsigned q0[4], q1[4], q2[4];
        split3x2(v[q][0].x, v[q][0].y, q0[0], q1[0], q2[0]);
        split3x2(v[q][0].z, v[q][0].w, q0[1], q1[1], q2[1]);
        split3x2(v[q][1].x, v[q][1].y, q0[2], q1[2], q2[2]);
        split3x2(v[q][1].z, v[q][1].w, q0[3], q1[3], q2[3]);
        const u32x4 p0 = {q0[0], q0[1], q0[2], q0[3]}, p1 = {q1[0], q1[1], q1[2], q1[3]}, p2 = {q2[0], q2[1], q2[2], q2[3]};
        unsigned char* dst = sA + kg * Cf::KG_BYTES + px * 16;
        *reinterpret_cast<u32x4*>(dst) = p0;
        *reinterpret_cast<u32x4*>(dst + Cf::PIECE_BYTES) = p1;
        *reinterpret_cast<u32x4*>(dst + 2 * Cf::PIECE_BYTES) = p2;
      }
    }
    store_w(wr, 0);
    __syncthreads();

#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) load_w(wr, kc, tap + 1);
      const int toff = ((tap / 3) - 1) * Cf::PW + (tap % 3) - 1;
      const unsigned char* wb = sW + (tap & 1) * Cf::WTAP_BYTES + g * Cf::WKG_BYTES + (wn * NT * 32 + li) * 16;
      u32x4 a[MT][3], bb[NT][3];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          a[mt][s] = *reinterpret_cast<const u32x4*>(sA + s * Cf::PIECE_BYTES + g * Cf::KG_BYTES + (lp[mt] + toff) * 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 3; ++s) bb[nt][s] = *reinterpret_cast<const u32x4*>(wb + s * Cf::WPIECE_BYTES + nt * 32 * 16);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          f32x16 c = acc[mt][nt];
          c = mfma(a[mt][2], bb[nt][0], c);  // smallest terms first
          c = mfma(a[mt][0], bb[nt][2], c);
          c = mfma(a[mt][1], bb[nt][1], c);
          c = mfma(a[mt][1], bb[nt][0], c);
          c = mfma(a[mt][0], bb[nt][1], c);
          c = mfma(a[mt][0], bb[nt][0], c);
          acc[mt][nt] = c;
        }
      if (tap + 1 < 9) {
        store_w(wr, (tap + 1) & 1);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: C/D layout of v_mfma_f32_32x32x16_bf16: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = nb * BN + (wn * NT + nt) * 32 + li;
    const float bv = bias[co];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        const long long m = m0 + (wm * MT + mt) * 32 + row;
        if (m < M) {
          float vv = acc[mt][nt][r] + bv;
          if (RES) vv = vv + res[m * C + co];
          out[m * C + co] = fmaxf(vv, 0.f);
        }
      }
    }
  }
}

template <class Cf>
static int launch(const float* x, const void* wp, const float* bias, const float* res, float* out, int B, hipStream_t stream) {
  static PerDeviceOnce attr[2];
  const int dev_ = PerDeviceOnce::device();
  const int which = res ? 1 : 0;
  if (!attr[which].done(dev_)) {
    const void* fn = res ? reinterpret_cast<const void*>(&conv3x3_kernel<Cf, true>) : reinterpret_cast<const void*>(&conv3x3_kernel<Cf, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr[which].set(dev_);
  }
  const long long M = (long long)B * Cf::S * Cf::S;
  const dim3 grid((unsigned)((M + Cf::BM - 1) / Cf::BM), Cf::NB);
  if (res) hipLaunchKernelGGL((conv3x3_kernel<Cf, true>), grid, dim3(HDN_BLOCK), Cf::LDS_BYTES, stream, x, (const u32x4*)wp, bias, res, out, B);
  else hipLaunchKernelGGL((conv3x3_kernel<Cf, false>), grid, dim3(HDN_BLOCK), Cf::LDS_BYTES, stream, x, (const u32x4*)wp, bias, res, out, B);
  return launch_status();
}

}  // namespace cv
}  // namespace hdn

//            S    C   WM WN MT NT
using CV_L1 = hdn::cv::Cfg<32, 64, 4, 1, 2, 2>;    // 256 pixels (8 rows) x 64 channels
using CV_L2 = hdn::cv::Cfg<16, 128, 4, 1, 1, 2>;   // 128 pixels (8 rows) x 64 channels
using CV_L3 = hdn::cv::Cfg<8, 256, 2, 2, 1, 1>;    // 64 pixels (one image) x 64 channels
using CV_L4 = hdn::cv::Cfg<4, 512, 1, 4, 1, 1>;    // 32 pixels (two images) x 128 channels

extern "C" int hdn_conv3x3_block_n(int S, int C) {
  if (S == 32 && C == 64) return CV_L1::BN;
  if (S == 16 && C == 128) return CV_L2::BN;
  if (S == 8 && C == 256) return CV_L3::BN;
  if (S == 4 && C == 512) return CV_L4::BN;
  return HDN_E_LIMIT;
}

extern "C" int hdn_conv3x3_bias_relu_f32(const float* x, const void* wpacked, const float* bias, const float* residual, float* out, int B, int S,
                                         int C, void* stream) {
  if (!x || !wpacked || !bias || !out) return HDN_E_NULL;
  if (B <= 0 || S <= 0 || C <= 0) return HDN_E_SHAPE;
  if (out == x) return HDN_E_ALIAS;  // (out == residual is fine: each element is read before it is written, by the same lane)
  if ((long long)B * S * S * C > 0x7fffffffLL) return HDN_E_LIMIT;
  if (!hdn::aligned16(x) || !hdn::aligned16(wpacked)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (S == 32 && C == 64) return hdn::cv::launch<CV_L1>(x, wpacked, bias, residual, out, B, s);
  if (S == 16 && C == 128) return hdn::cv::launch<CV_L2>(x, wpacked, bias, residual, out, B, s);
  if (S == 8 && C == 256) return hdn::cv::launch<CV_L3>(x, wpacked, bias, residual, out, B, s);
  if (S == 4 && C == 512) return hdn::cv::launch<CV_L4>(x, wpacked, bias, residual, out, B, s);
  return HDN_E_LIMIT;
}
