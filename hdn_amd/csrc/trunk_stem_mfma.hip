// First stage of the homography regressor's ResNet-34 trunk on the matrix cores (round 5; the batches that fill the chip):
//   y = maxpool3x3/s2/p1( relu( conv7x7/s2/p3(x, w) + b ) ),   x [B,2,127,127] (NCHW) -> y [B,32,32,64] (channels-last)
// Reference: HomoResNet.forward, homo_estimator/Deep_homography/Oneline_DLTv1/backbone/resnet.py:141-147,183-186 (conv1 / bn1 / relu / maxpool,
// eval mode, BatchNorm folded into (w, b) by the host).  trunk_stem.hip does the same on the vector pipe (53 us at B = 64: packed-FMA issue, 38 % of the
// fp32 peak) and stays the form for small batches and other sizes; this one is the implicit GEMM
//   D[conv pixel][co] = sum over k = (ci, ky, kx) of A[pixel][k] * W[k][co],      K = 2 x 7 x 8 = 112 (kx padded to 8 with a zero weight)
// with fp32 carried as two fp16 pieces and three piece products into hi / lo accumulators (conv3x3.hip has the error analysis: the error of an fp32
// convolution).  The k order makes an MFMA A fragment (lane = (pixel, k half), 8 consecutive k) EIGHT CONSECUTIVE INPUT PIXELS of one input row:
// k = 16 s + 8 g + j  <->  row r = 2 s + g = ci * 7 + ky,  kx = j,  so lane (ox, g) of k step s needs x[ci][2 oy + ky - 3][2 ox - 3 + j], j = 0..7.
//
// Workgroup = (image, ROWS conv rows).  Its 2 ROWS + 7 input rows are split ONCE into the two fp16 pieces while they are staged (LDS images
// [ci][piece][row][192 halves], zero-padded: 3 columns left, rows outside the image as zeros - no masking later); an A fragment is then four dwords at a
// 4-byte-aligned address (two ds_read2_b32), no arithmetic in the loop.  The row pitch of 96 dwords puts the two k halves of a wave (consecutive rows) on
// disjoint LDS banks.  Weights: host-packed in fragment order, copied to LDS once.
// Wave = (channel half h, stream of TILES tiles); a tile = 2 conv rows x 64 columns x 32 channels = 4 x 1 MFMA tiles, 7 k steps x 12 MFMAs.  The
// 3 x 3 / stride-2 max pool never leaves the registers: rows 2p and 2p + 1 of a tile sit in the same lane / register of two accumulators, row 2p - 1 is
// the carry from the stream's previous tile (a stream starts with the second row of the tile above its first: 6 MFMAs per k step), and in the 32 x 32
// C / D layout a lane holds 4 consecutive columns - the one column to the left comes from lane ^ 32 (one ds_bpermute per 4 columns).  Everything is
// >= 0 behind the ReLU and every window holds a real element, so the pool's -inf padding is a plain skip (a zero carry / zero left neighbour).
// Two waves per SIMD (8 per workgroup at ROWS = 16): one wave's epilogue runs under the other's MFMAs.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "hdn_common.h"
#include "mfma_split.h"

namespace hdn {
namespace stem_mc {
using namespace hdn::mc;

constexpr int H = 127, W = 127, HC = 64, HP = 32, WP = 32, CO = 64, NSTEP = 7;
constexpr int PAIRS = 68;                                      // staged pairs of padded columns per row: 3 zeros + 127 pixels + 6 zeros
constexpr int ROW_BYTES = 384;                                 // 96 dwords: consecutive rows 32 banks apart
constexpr int W_WORDS = NSTEP * 2 * 2 * 64;                    // 16-byte words: [k step][n tile][piece][lane]
constexpr int W_BYTES = W_WORDS * 16;

template <int ROWS, int STREAMS>
struct Geo {
  static constexpr int THREADS = 128 * STREAMS;                // 2 channel halves x STREAMS waves
  static constexpr int NBLK = HC / ROWS;                       // workgroups per image
  static constexpr int NROWS = 2 * ROWS + 7;                   // input rows of ROWS conv rows + the conv row above them
  static constexpr int TILES = ROWS / 2 / STREAMS;             // tiles (pooled rows) per stream
  static constexpr int PIECE_BYTES = NROWS * ROW_BYTES;
  static constexpr int A_BYTES = 4 * PIECE_BYTES;              // [ci][piece]
  static constexpr int LDS_BYTES = W_BYTES + A_BYTES;
  static constexpr int ITEMS = NROWS * 2 * PAIRS;
  static_assert(ROWS % (2 * STREAMS) == 0 && HC % ROWS == 0, "geometry");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct __attribute__((packed, aligned(4))) Frag {              // 8 halves at a 4-byte-aligned LDS address
  unsigned d[4];
};

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int ROWS, int STREAMS, bool OUT_SD = false>
__global__ __launch_bounds__(128 * STREAMS) void trunk_stem_mfma_kernel(const float* __restrict__ x, const u32x4* __restrict__ wfrag,
                                                                        const float* __restrict__ bias, float* __restrict__ out) {
  using G = Geo<ROWS, STREAMS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* const sW = reinterpret_cast<u32x4*>(smem);
  unsigned char* const sA = smem + W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave & 1, rr = wave >> 1;
  const int li = lane & 31, g = lane >> 5;
  const int b = blockIdx.x / G::NBLK, q = blockIdx.x % G::NBLK;
  const int R0 = q * ROWS;                                     // first conv row of the workgroup
  const float* const xb = x + (size_t)b * 2 * H * W;

  // ---- weights -> LDS; input rows 2 R0 - 5 .. 2 R0 + 2 ROWS + 1 -> fp16 pieces in LDS (every load in flight before the first use)
  for (int i = tid; i < W_WORDS; i += G::THREADS) sW[i] = wfrag[i];
  {
    constexpr int NIT = cdiv(G::ITEMS, G::THREADS);
    f2 v[NIT];
    const int y_lo = 2 * R0 - 5;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int item = min(tid + k * G::THREADS, G::ITEMS - 1);
      const int pr = item % PAIRS, rc = item / PAIRS, ci = rc & 1, y = y_lo + (rc >> 1);
      const int x0 = 2 * pr - 3;                               // image column of the pair's first element
      const float* src = xb + ((size_t)ci * H + min(max(y, 0), H - 1)) * W;
      const float a = src[min(max(x0, 0), W - 1)], c = src[min(max(x0 + 1, 0), W - 1)];
      const bool yin = y >= 0 && y < H;
      v[k] = f2{yin && x0 >= 0 && x0 < W ? a : 0.f, yin && x0 + 1 >= 0 && x0 + 1 < W ? c : 0.f};
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int item = tid + k * G::THREADS;
      if (item < G::ITEMS) {
        const int pr = item % PAIRS, rc = item / PAIRS, ci = rc & 1, slot = rc >> 1;
        unsigned p0, p1;
        split2(v[k], p0, p1);
        unsigned char* dst = sA + (size_t)(ci * 2) * G::PIECE_BYTES + slot * ROW_BYTES + pr * 4;
        *reinterpret_cast<unsigned*>(dst) = p0;
        *reinterpret_cast<unsigned*>(dst + G::PIECE_BYTES) = p1;
      }
    }
  }
  const float bias_c = bias[32 * h + li];                      // C / D layout: column (channel) = lane & 31
  __syncthreads();

  // ---- per-lane A address pieces: k step s, half g -> r = 2 s + g = ci * 7 + ky; input row slot of conv row oy, tap ky = 2 (oy - R0) + ky + 2
  const int Rw = R0 + rr * 2 * G::TILES;                       // the stream's first conv row
  const unsigned char* const a_lane = sA + 4 * li + (2 * (Rw - R0)) * ROW_BYTES;
  f32x16 carry[2];                                             // conv row above the current tile, [column half], after the ReLU
  f32x16 acc[4], accl[4];                                      // [row of the pair * 2 + column half]

  auto tile = [&](auto HaloC, int t) {
    constexpr bool HALO = decltype(HaloC)::value;              // only the second row of the pair (the stream's carry-in)
    constexpr int M0 = HALO ? 2 : 0;
    const unsigned char* const a_tile = a_lane + (4 * t + 2) * ROW_BYTES;
#pragma unroll
    for (int mt = M0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mt][r] = bias_c * ACT_SCALE;      // (the accumulators hold sums of activations x 2^-8: join() gives the bias back exactly)
        accl[mt][r] = 0.f;
      }
    static_for<NSTEP>([&](auto Sc) {
      constexpr int s = decltype(Sc)::value;
      constexpr int r0 = 2 * s, r1 = 2 * s + 1;
      constexpr int o0 = ((r0 / 7) * 2 * G::NROWS + (r0 % 7)) * ROW_BYTES, o1 = ((r1 / 7) * 2 * G::NROWS + (r1 % 7)) * ROW_BYTES;
      const unsigned char* const pa = a_tile + (g ? o1 : o0);
      u32x4 a[4][2], bf[2];
#pragma unroll
      for (int mt = M0; mt < 4; ++mt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const Frag f = *reinterpret_cast<const Frag*>(pa + pc * G::PIECE_BYTES + (mt >> 1) * 2 * ROW_BYTES + (mt & 1) * 128);
          a[mt][pc] = u32x4{f.d[0], f.d[1], f.d[2], f.d[3]};
        }
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) bf[pc] = sW[((s * 2 + h) * 2 + pc) * 64 + lane];
#pragma unroll
      for (int mt = M0; mt < 4; ++mt) accl[mt] = mfma(a[mt][1], bf[0], accl[mt]);
#pragma unroll
      for (int mt = M0; mt < 4; ++mt) acc[mt] = mfma(a[mt][0], bf[0], acc[mt]);
#pragma unroll
      for (int mt = M0; mt < 4; ++mt) accl[mt] = mfma(a[mt][0], bf[1], accl[mt]);
    });
    if constexpr (HALO) {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int r = 0; r < 16; ++r) carry[ch][r] = fmaxf(join<OUT_SD>(acc[2 + ch][r], accl[2 + ch][r]), 0.f);
    } else {
      // rows 2p - 1 (carry), 2p, 2p + 1 -> vertical max; then columns 2 px - 1 .. 2 px + 1.  Lane (li, g) holds columns 32 ch + 8 j + 4 g + (0..3).
      const int p = (Rw >> 1) + t;
      float* const orow = out + (((size_t)b * HP + p) * WP + 2 * g) * CO + 32 * h + li;
      float left0 = 0.f;                                       // column 31's value for column-half 1, j = 0 (lanes g = 0)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        float v[16], tl[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v0 = join<OUT_SD>(acc[ch][r], accl[ch][r]), v1 = join<OUT_SD>(acc[2 + ch][r], accl[2 + ch][r]);
          v[r] = max3(v0, v1, carry[ch][r]);                   // (carry >= 0: the ReLU of all three)
          carry[ch][r] = fmaxf(v1, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tl[j] = __shfl_xor(v[4 * j + 3], 32);          // the other k half's last column of group j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float left = g ? tl[j] : (j > 0 ? tl[j - 1] : left0);               // column 32 ch + 8 j + 4 g - 1 (image column -1: 0)
          float* const o = orow + (size_t)(16 * ch + 4 * j) * CO;
          o[0] = max3(left, v[4 * j], v[4 * j + 1]);
          o[CO] = max3(v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        }
        left0 = tl[3];
      }
    }
  };

  if (Rw > 0) {
    tile(std::true_type{}, -1);
  } else {
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int r = 0; r < 16; ++r) carry[ch][r] = 0.f;
  }
#pragma unroll 1
  for (int t = 0; t < G::TILES; ++t) tile(std::false_type{}, t);
}

template <int ROWS, int STREAMS, bool OUT_SD = false>
int launch(const float* x, const void* wfrag, const float* bias, float* out, int B, hipStream_t stream) {
  using G = Geo<ROWS, STREAMS>;
  static PerDeviceOnce attr;
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trunk_stem_mfma_kernel<ROWS, STREAMS, OUT_SD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::LDS_BYTES);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr.set(dev_);
  }
  hipLaunchKernelGGL((trunk_stem_mfma_kernel<ROWS, STREAMS, OUT_SD>), dim3((unsigned)B * G::NBLK), dim3(G::THREADS), G::LDS_BYTES, stream, x,
                     static_cast<const u32x4*>(wfrag), bias, out);
  return launch_status();
}

}  // namespace stem_mc
}  // namespace hdn

// wfrag: [7 k steps][2 n tiles][2 pieces][64 lanes = k half x 32 + n][8] fp16 (hdn_amd.trunk.pack_stem_mfma): element j of lane (g, n) of k step s is
// piece pc of w[co = tile * 32 + n][ci][ky][kx = j] with ci * 7 + ky = 2 s + g, and 0 for j = 7.
extern "C" int hdn_trunk_stem_mfma_f32(const float* x, const void* wfrag, const float* bias, float* out, int B, int H, int W, int out_domain,
                                       void* stream) {
  if (!x || !wfrag || !bias || !out) return HDN_E_NULL;
  if (B <= 0 || (out_domain != 0 && out_domain != 1)) return HDN_E_SHAPE;
  if (H != hdn::stem_mc::H || W != hdn::stem_mc::W || B > (1 << 20)) return HDN_E_LIMIT;       // 127-px crops only; hdn_trunk_stem_f32 takes the rest
  if (static_cast<const void*>(out) == static_cast<const void*>(x)) return HDN_E_ALIAS;
  if (!hdn::aligned16(wfrag) || !hdn::aligned16(out)) return HDN_E_LIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = hdn::check_fp16_range(x, (long long)B * 2 * H * W, st)) return rc;
  static const int rows_env = [] {
    const char* e = getenv("HDN_STEM_ROWS");                   // A/B switch (tools/experiments): conv rows per workgroup
    return e ? atoi(e) : 0;
  }();
  // 4 workgroups per image from 48 images on, 8 from 8 on, 16 for the tracker's few-image calls (B = 1, cold: 8.8 us against 11.7 at 8 rows, 12.0 on
  // the vector pipe): the shorter the workgroup's chain of dependent steps the better when nothing else is on the chip
  const int rows = rows_env ? rows_env : (B >= 48 ? 16 : B >= 8 ? 8 : 4);
  // out_domain 1: the pooled output is left as relu(conv) x 2^-8 (the input is split as x 2^-8 either way; only the final multiply by 2^8 is dropped)
  if (out_domain) {
    if (rows == 16) return hdn::stem_mc::launch<16, 4, true>(x, wfrag, bias, out, B, st);
    if (rows == 4) return hdn::stem_mc::launch<4, 2, true>(x, wfrag, bias, out, B, st);
    return hdn::stem_mc::launch<8, 2, true>(x, wfrag, bias, out, B, st);
  }
  if (rows == 16) return hdn::stem_mc::launch<16, 4>(x, wfrag, bias, out, B, st);
  if (rows == 4) return hdn::stem_mc::launch<4, 2>(x, wfrag, bias, out, B, st);
  return hdn::stem_mc::launch<8, 2>(x, wfrag, bias, out, B, st);
}
