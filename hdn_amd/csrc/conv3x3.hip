// 3x3 / stride 1 / pad 1 convolution + bias + (residual) + ReLU of the homography regressor's trunk as ONE implicit-GEMM kernel
// on the matrix cores.  Replaces, per BasicBlock of
// homo_estimator/Deep_homography/Oneline_DLTv1/backbone/resnet.py:78-94 (eval mode, BatchNorm folded into the weights), a MIOpen
// fp32 convolution plus the elementwise tail.  29 of the trunk's 36 convolutions have this shape (C -> C channels at S x S,
// (C, S) = (64, 32), (128, 16), (256, 8), (512, 4) for 127-px crops).
//
// GEMM view (NHWC): D[pixel][cout] = sum over (tap, cin) A_tap[pixel][cin] * W[tap][cin][cout], A_tap = the input shifted by the tap.
// fp32 in, fp32 out, but the products run on the 16-bit matrix pipe at 16x the fp32 MFMA rate.  Round 4: every fp32 value is split
// into TWO fp16 pieces, x = h0 + 2^-11 h1 with h0 = fp16(x) and h1 = fp16((x - h0) * 2^11) (the residual is exact in fp32; scaling
// it by 2^11 keeps it in fp16's normal range whenever x is: 11 + 11 significand bits, |x - h0 - 2^-11 h1| <= 2^-23 |x|), and THREE
// piece products are accumulated in fp32 in two accumulator sets: hi += x0 w0, lo += x0 w1 + x1 w0, result = hi + 2^-11 lo (the
// dropped x1 w1 is below 2^-22 of the product).  3 x 32 clk per K = 16 against 8 x 64 clk on v_mfma_f32_32x32x2_f32 and against
// the 6 x 32 clk of round 3's three-bf16-piece form; measured against float64 the result has the error of an fp32 convolution
// (rms 6.6e-8 of the output scale on K = 2,304 sums, fp32 sgemm: 6.6e-8, the bf16 form: 2.8e-8).  Range: |x| < 1.67e7 (activations are split as x 2^-8, mfma_split.h), weights < 65,504 (fp16);
// the trunk's activations (BatchNorm-folded, ReLU) are O(10).  The weights are split once on the host
// (hdn_amd.trunk.pack_conv3x3), the activations while they are staged into LDS.
//
// Workgroup = 8 waves (4 consumers issuing MFMAs + 4 producers staging operands, see conv3x3_kernel), tile = BM output pixels
// (consecutive in (b, y, x) order: whole image rows) x BN output channels.
//   LDS A image: the tile's input pixels with a one-pixel halo (zeros outside the image), one K chunk of 16 * KS input channels at
//                a time, as [piece][k step][k half][pixel] x 16 B: an MFMA A fragment (lane = (pixel row i, k half g), 8 fp16) is
//                one ds_read_b128 and a tap is a constant address offset; conflict-free through the row / image pitches of Cfg and
//                the lane -> pixel order of mrow_to_pixel().  Two images (double buffer).
//   LDS W image: [tap of the stage][k step][piece][k half][cout] x 16 B, one STAGE = one kernel row (3 taps) of one chunk, a ring
//                of three stages, streamed from the host-packed layout (which is exactly this order).
//   pipeline   : the producers run two stages ahead of the consumers (weights in registers two more stages ahead of that); the
//                consumers read a step's fragments one step ahead, across stage boundaries too: one barrier per stage.
//   wave tile  : MT x NT MFMA tiles of 32 x 32; accumulators stay in registers over the whole K loop;
//   epilogue   : the tile goes through LDS once, then + bias[cout] (+ residual) -> ReLU -> NHWC store, 16 bytes per lane, by all 512 threads.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "hdn_common.h"
#include "mfma_split.h"

// Measurement hooks (tools/build_variant.sh ... -DHDN_ABLATION -D<experiment>): every site below expands to its production text; the
// experiments' replacement bodies live in ablation/conv3x3.inc and are compiled in only under -DHDN_ABLATION, so that editing or adding an
// experiment leaves this translation unit's text (and the hash the committed PMC record carries) unchanged.
#define HDN_ABL_CONV3X3_0(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_1(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_2(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_3(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_4(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_5(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_6(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_7(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_8(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_9(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_10(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_11(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_12(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_13(...) __VA_ARGS__
#define HDN_ABL_CONV3X3_14_BEGIN
#define HDN_ABL_CONV3X3_14_END
#ifdef HDN_ABLATION
#include "ablation/conv3x3.inc"
#endif

namespace hdn {
namespace cv {
using namespace hdn::mc;

// vector types, the MFMA, the fp32 -> two-fp16-piece split (x = p0 + 2^-11 p1) and static_for: mfma_split.h

// S = OUTPUT side, CI -> CO channels, STRIDE 1 or 2 (input side S * STRIDE); DS: the block's 1x1 / stride-2 downsample branch is
// computed alongside from the same staged activations (it is the centre tap with its own weights) into a second output.
template <int S_, int CI_, int CO_, int STRIDE_, bool DS_, int WM_, int WN_, int MT_, int NT_, int KS_>
struct Cfg {
  static constexpr int S = S_, CI = CI_, CO = CO_, STRIDE = STRIDE_, WM = WM_, WN = WN_, MT = MT_, NT = NT_, KS = KS_;
  static constexpr bool DS = DS_;
  static constexpr int SI = S * STRIDE;
  static constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(STRIDE == 1 || STRIDE == 2, "stride");
  static_assert(CO % BN == 0 && CI % (16 * KS) == 0, "channel blocking");
  static_assert((BM % S == 0) && ((S * S) % BM == 0 || BM % (S * S) == 0), "a tile is whole rows of one image, or whole images");
  static constexpr int IMGS = BM > S * S ? BM / (S * S) : 1;      // images per tile
  static constexpr int R = BM / (S * IMGS);                      // output rows per image in the tile
  static constexpr int PWV = SI + 2, PH = STRIDE == 1 ? R + 2 : 2 * R + 1;  // halo'ed input patch
  // Pitches of the LDS image.  A ds_read_b128 is served in groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} of a half
  // wave), which must fall on 16 different 16-byte columns of the 256-byte bank row.  With image rows shorter than 16 pixels that
  // takes a row pitch of 12 pixels at S = 8 and an image pitch of 40 at S = 4, together with the lane -> pixel order of
  // mrow_to_pixel() below (40 % of the LDS cycles of those stages were bank conflicts before).
  static constexpr int PW = (STRIDE == 1 && S == 8) ? 12 : PWV;
  static constexpr int IPITCH = PH * PW + ((STRIDE == 1 && S == 4) ? 4 : 0);
  // ... and the producers' ds_write_b128 (8 consecutive lanes = 8 / (2 KS) pixels x the 2 KS (k step, k half) sub-images of a pixel, one
  // 128-byte bank row) needs the sub-images 128 / (2 KS) bytes apart modulo 128: LP = 8 / (2 KS) modulo 8.
  static constexpr int LPV = IMGS * IPITCH;                      // LDS pixels that exist
  static constexpr int LP = LPV + ((8 / (2 * KS) - LPV % 8) + 8) % 8;
  static constexpr int NP = 2;                                   // fp16 pieces per value
  static constexpr int KG_BYTES = LP * 16, KSTEP_BYTES = 2 * KG_BYTES, PIECE_BYTES = KS * KSTEP_BYTES, A_BYTES = NP * PIECE_BYTES;
  static constexpr int WKG_BYTES = BN * 16, WPIECE_BYTES = 2 * WKG_BYTES, WSTEP_BYTES = NP * WPIECE_BYTES;   // one (tap, k step)
  static constexpr int NTAP = DS ? 4 : 3;                        // taps of a stage: a kernel row (+ the downsample tap, used in the middle row)
  static constexpr int WSTAGE_BYTES = NTAP * KS * WSTEP_BYTES;   // one kernel row of one chunk
  static constexpr int EPI_STRIDE = BN + 4;                      // floats per pixel row of the output staging (pad: bank spread of the two half waves)
  static constexpr int EPI_BYTES = BM * EPI_STRIDE * 4;
  static constexpr int LDS_BYTES = (2 * A_BYTES + 3 * WSTAGE_BYTES) > EPI_BYTES ? (2 * A_BYTES + 3 * WSTAGE_BYTES) : EPI_BYTES;   // two A images, a ring of three W stages
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the LDS");
  static constexpr int NCHUNK = CI / (16 * KS), NB = CO / BN, NSTAGE = 3 * NCHUNK;
  static constexpr int W4 = WSTAGE_BYTES / 16;                   // 16-byte words of one stage's weights
  static constexpr int WITER = cdiv(W4, HDN_BLOCK);
  static constexpr int AITEMS = LP * 2 * KS, AITER = cdiv(AITEMS, HDN_BLOCK);   // (pixel, k step, k half) items of 8 channels
};

// Row i (0..31) of an MFMA tile -> pixel of the tile's 32-pixel block, (image, y, x) order.  Stride-1 tiles with rows shorter than
// 32 pixels hand the two 16-lane access groups of a fragment read pixel sets that are 16 distinct columns of the bank row:
// group = which of the two, k = rank inside it; S = 16: group = row; S = 8: rows (0, 2) | (1, 3); S = 4: even | odd rows of two images.
template <int S>
__host__ __device__ constexpr __forceinline__ int mrow_to_pixel_s1(int i) {   // stride-1 tiles of side S
  if constexpr (S >= 32) {
    return i;
  } else {
    const int q = i >> 2, grp = (0x96 >> q) & 1, k = ((q >> 1) << 2) | (i & 3);
    if constexpr (S == 16) return grp * 16 + k;
    else if constexpr (S == 8) return (2 * (k >> 3) + grp) * 8 + (k & 7);
    else return (k >> 3) * 16 + (2 * ((k >> 2) & 1) + grp) * 4 + (k & 3);
  }
}
template <class Cf>
__host__ __device__ constexpr __forceinline__ int mrow_to_pixel(int i) {
  if constexpr (Cf::STRIDE != 1) return i;
  else return mrow_to_pixel_s1<Cf::S>(i);
}

// A convolution input that was never written out as an activation (B = 1: every launch is a dependent step of ~5 us, and the
// launch that only adds the K slices up was half of them): the raw K-slice sums [zx][B][SI][SI][CI] of the convolution that
// produced it, finished while it is staged: relu(sum of the slices in slice order + bias (+ residual)), the arithmetic of
// conv3x3_reduce_kernel.  The residual is an activation (zr = 1) or, behind a downsample branch, raw slices again.  `xout`: the
// finished activation is also written out (the NEXT block's residual) — every element by exactly one workgroup (output-channel
// block 0; the tile's own rows; the K slice's channels).
struct LazyIn {
  const float* bias;
  const float* res;
  float* xout;
  int zx, zr;
};

// s (+)= the N slices p[j * MX] (8 floats each), all N loads in flight at once, added in slice order
template <int N>
__device__ __forceinline__ void add_slices(const float* p, size_t MX, bool init, f4& s0, f4& s1) {
  f4 a[N][2];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const f4* q = reinterpret_cast<const f4*>(p + (size_t)j * MX);
    a[j][0] = q[0];
    a[j][1] = q[1];
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (init && j == 0) s0 = a[0][0], s1 = a[0][1];
    else s0 = s0 + a[j][0], s1 = s1 + a[j][1];
  }
}
__device__ __forceinline__ void sum_slices(const float* p, size_t MX, int z, f4& s0, f4& s1) {
  int j = 0;
  for (; j + 16 <= z; j += 16) add_slices<16>(p + (size_t)j * MX, MX, j == 0, s0, s1);
  if (z & 8) { add_slices<8>(p + (size_t)j * MX, MX, j == 0, s0, s1); j += 8; }
  if (z & 4) { add_slices<4>(p + (size_t)j * MX, MX, j == 0, s0, s1); j += 4; }
  if (z & 2) { add_slices<2>(p + (size_t)j * MX, MX, j == 0, s0, s1); j += 2; }
  if (z & 1) add_slices<1>(p + (size_t)j * MX, MX, j == 0, s0, s1);
}

// MODE 0: out = relu(conv + bias); 1: out = relu(conv + bias + res); 2: the K slice blockIdx.z of `cps` chunks, raw sums to
// out[blockIdx.z][M][CO] (the workspace; conv3x3_reduce_kernel finishes).  Cf::DS: out2 = the downsample branch (raw sums, no
// bias / ReLU; mode 2: out2[blockIdx.z][M][CO]).
// 8 waves, two per SIMD, with separate roles: waves 0-3 only read fragments and issue MFMAs (consumers), waves 4-7 only load from
// global memory, split to fp16 pieces and fill the LDS images (producers).  A consumer never waits for a global load or for the operands
// of an LDS store; the two roles meet at the one barrier per stage, and the producers run TWO stages ahead, so that a consumer can
// read the first fragments of stage s + 1 while it still issues the MFMAs of stage s.
// LAZY: x is a LazyIn's slices (mode 2 only: the chained form of the small-batch trunk, launch_chain).
template <class Cf, int MODE, bool LAZY = false, bool SD = false>
__global__ __launch_bounds__(2 * HDN_BLOCK) void conv3x3_kernel(const float* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                            const float* __restrict__ res, float* __restrict__ out, float* __restrict__ out2, int B,
                                                            int cps, LazyIn lz) {
  static_assert(!LAZY || MODE == 2, "a lazy input feeds the K-sliced form");
  constexpr bool RES = MODE == 1, PARTIAL = MODE == 2, DS = Cf::DS;
  constexpr int S = Cf::S, SI = Cf::SI, CI = Cf::CI, C = Cf::CO, MT = Cf::MT, NT = Cf::NT, BM = Cf::BM, BN = Cf::BN, KS = Cf::KS, ST = Cf::STRIDE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sA = smem;
  unsigned char* const sW = smem + 2 * Cf::A_BYTES;
  const int tid = threadIdx.x & (HDN_BLOCK - 1), lane = tid & 63;   // index inside the role (or the workgroup)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool produce = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0, consume = !produce;
  const int wm = wave / Cf::WN, wn = wave % Cf::WN;
  const int li = lane & 31, g = lane >> 5;
  const long long m0 = (long long)blockIdx.x * BM;          // first output pixel of the tile, (b, y, x) order
  const int nb = blockIdx.y;                                 // output-channel block
  const int b0 = (int)(m0 / (S * S)), y0 = (int)((m0 % (S * S)) / S);
  const long long M = (long long)B * S * S;

  // ---- per-lane LDS byte offsets: A fragments of this wave's M tiles (centre tap), B fragments of its N tiles
  uint32_t aoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int i = (wm * MT + mt) * 32 + mrow_to_pixel<Cf>(li);   // pixel inside the tile
    const int img = i / (Cf::R * S), yy = (i / S) % Cf::R, xx = i % S;
    aoff[mt] = lds_addr(sA) + g * Cf::KG_BYTES + (img * Cf::IPITCH + (ST * yy + 1) * Cf::PW + (ST * xx + 1)) * 16;
  }
  const uint32_t boff = lds_addr(sW) + g * Cf::WKG_BYTES + (wn * NT * 32 + li) * 16;

  f32x16 acc[MT][NT], accl[MT][NT];        // hi: x0 w0; lo: x0 w1 + x1 w0 (scaled by 2^11)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = accl[mt][nt][r] = 0.f;
  f32x16 accd[DS ? MT : 1][DS ? NT : 1], accdl[DS ? MT : 1][DS ? NT : 1];   // the downsample branch
  if (DS) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accd[mt][nt][r] = accdl[mt][nt][r] = 0.f;
  }

  // this channel block's packed weights: [chunk][kernel row][tap in row][k step][piece][k half][n][8 fp16] = [stage][W4 words]
  const int chunk0 = PARTIAL ? (int)blockIdx.z * cps : 0, nchunk = PARTIAL ? cps : Cf::NCHUNK, nstage = 3 * nchunk;   // this launch's K range
  const u32x4* wblock = wp + ((size_t)nb * Cf::NSTAGE + (size_t)chunk0 * 3) * Cf::W4;
  u32x4 wr[2][Cf::WITER];   // two stages in flight: stage t travels in wr[t & 1] from stage t - 4 (load) to stage t - 2 (LDS store)
  auto load_w = [&](int stage, auto P) {
    constexpr int p = decltype(P)::value;
    HDN_ABL_CONV3X3_0()
    const u32x4* src = wblock + (size_t)stage * Cf::W4;
#pragma unroll
    for (int q = 0; q < Cf::WITER; ++q) wr[p][q] = src[min(tid + q * HDN_BLOCK, Cf::W4 - 1)];
  };
  auto store_w = [&](auto P, int buf) {
    constexpr int p = decltype(P)::value;
    u32x4* dst = reinterpret_cast<u32x4*>(sW + buf * Cf::WSTAGE_BYTES);
#pragma unroll
    for (int q = 0; q < Cf::WITER; ++q)
      if (tid + q * HDN_BLOCK < Cf::W4) dst[tid + q * HDN_BLOCK] = wr[p][q];
  };
  // input chunk: (pixel, k step, k half) items of 8 channels = 32 bytes
  f4 av[Cf::AITER][2];
  auto load_a = [&](int chunk) {
    HDN_ABL_CONV3X3_1()
#pragma unroll
    for (int q = 0; q < Cf::AITER; ++q) {
      const int item = tid + q * HDN_BLOCK;
      const int px = min(item / (2 * KS), Cf::LP - 1), sub = item % (2 * KS);
      const int img = px / Cf::IPITCH, ry = (px % Cf::IPITCH) / Cf::PW, rx = px % Cf::IPITCH % Cf::PW;   // (pad pixels: zeros)
      const int b = b0 + img, y = ST * y0 + ry - 1, xx = rx - 1;
      const bool ok = item < Cf::AITEMS && img < Cf::IMGS && b < B && ry < Cf::PH && y >= 0 && y < SI && xx >= 0 && xx < SI;
      const f4* src = reinterpret_cast<const f4*>(x + (((size_t)b * SI + y) * SI + xx) * CI + (chunk0 + chunk) * (16 * KS) + sub * 8);
      av[q][0] = ok ? src[0] : f4{0.f, 0.f, 0.f, 0.f};
      av[q][1] = ok ? src[1] : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_a = [&](int ab) {
    HDN_ABL_CONV3X3_2()
#pragma unroll
    for (int q = 0; q < Cf::AITER; ++q) {
      const int item = tid + q * HDN_BLOCK;
      if (item < Cf::AITEMS) {
        const int px = item / (2 * KS), sub = item % (2 * KS);   // sub = k step * 2 + k half
        unsigned q0[4], q1[4];
        split2x2<SD>(av[q][0].x, av[q][0].y, q0[0], q1[0]);
        split2x2<SD>(av[q][0].z, av[q][0].w, q0[1], q1[1]);
        split2x2<SD>(av[q][1].x, av[q][1].y, q0[2], q1[2]);
        split2x2<SD>(av[q][1].z, av[q][1].w, q0[3], q1[3]);
        unsigned char* dst = sA + ab * Cf::A_BYTES + sub * Cf::KG_BYTES + px * 16;
        *reinterpret_cast<u32x4*>(dst) = u32x4{q0[0], q0[1], q0[2], q0[3]};
        *reinterpret_cast<u32x4*>(dst + Cf::PIECE_BYTES) = u32x4{q1[0], q1[1], q1[2], q1[3]};
      }
    }
  };

  // LAZY: the finished input of this launch's (at most two) chunks goes straight into the two A images, by all 512 threads, before
  // the roles split: one round trip of zx (+ zr) slice loads per item.
  if constexpr (LAZY) {
    if (produce) {   // the first two weight stages travel meanwhile
      load_w(0, std::integral_constant<int, 0>{});
      load_w(1, std::integral_constant<int, 1>{});
    }
    const size_t MX = (size_t)B * SI * SI * CI;
    for (int it = threadIdx.x; it < nchunk * Cf::AITEMS; it += 2 * HDN_BLOCK) {
      const int chunk = it / Cf::AITEMS, item = it % Cf::AITEMS;
      const int px = item / (2 * KS), sub = item % (2 * KS);
      const int img = px / Cf::IPITCH, ry = (px % Cf::IPITCH) / Cf::PW, rx = px % Cf::IPITCH % Cf::PW;
      const int b = b0 + img, y = ST * y0 + ry - 1, xx = rx - 1;
      const bool ok = img < Cf::IMGS && b < B && ry < Cf::PH && y >= 0 && y < SI && xx >= 0 && xx < SI;
      f4 s0 = f4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
      if (ok) {
        const int ch = (chunk0 + chunk) * (16 * KS) + sub * 8;
        const size_t off = (((size_t)b * SI + y) * SI + xx) * CI + ch;
        const f4 b0v = *reinterpret_cast<const f4*>(lz.bias + ch), b1v = *reinterpret_cast<const f4*>(lz.bias + ch + 4);
        f4 r0 = s0, r1 = s0;
        if (lz.res) sum_slices(lz.res + off, MX, lz.zr, r0, r1);
        sum_slices(x + off, MX, lz.zx, s0, s1);
        s0 = s0 + b0v; s1 = s1 + b1v;
        if (lz.res) { s0 = s0 + r0; s1 = s1 + r1; }
        s0.x = fmaxf(s0.x, 0.f); s0.y = fmaxf(s0.y, 0.f); s0.z = fmaxf(s0.z, 0.f); s0.w = fmaxf(s0.w, 0.f);
        s1.x = fmaxf(s1.x, 0.f); s1.y = fmaxf(s1.y, 0.f); s1.z = fmaxf(s1.z, 0.f); s1.w = fmaxf(s1.w, 0.f);
        if (lz.xout && nb == 0 && ry >= 1 && ry <= Cf::PH - (ST == 1 ? 2 : 1)) {   // the tile's own input rows (not the halo)
          *reinterpret_cast<f4*>(lz.xout + off) = s0;
          *reinterpret_cast<f4*>(lz.xout + off + 4) = s1;
        }
      }
      unsigned q0[4], q1[4];
      split2x2<SD>(s0.x, s0.y, q0[0], q1[0]);
      split2x2<SD>(s0.z, s0.w, q0[1], q1[1]);
      split2x2<SD>(s1.x, s1.y, q0[2], q1[2]);
      split2x2<SD>(s1.z, s1.w, q0[3], q1[3]);
      unsigned char* dst = sA + chunk * Cf::A_BYTES + sub * Cf::KG_BYTES + px * 16;
      *reinterpret_cast<u32x4*>(dst) = u32x4{q0[0], q0[1], q0[2], q0[3]};
      *reinterpret_cast<u32x4*>(dst + Cf::PIECE_BYTES) = u32x4{q1[0], q1[1], q1[2], q1[3]};
    }
  }

  struct Frags {
    u32x4 a[MT][Cf::NP], b[NT][Cf::NP];
  };
  // fragments of step (tap t of the stage's kernel row ky, k step ks) from W buffer `buf`
  auto read_frags = [&](Frags& f, int ky, int t, int ks, int buf, int ab) {
    HDN_ABL_CONV3X3_3()
    const int toff = (t == 3 ? 0 : ((ky - 1) * Cf::PW + (t - 1)) * 16) + ks * Cf::KSTEP_BYTES + ab * Cf::A_BYTES;   // (tap 3: the downsample branch reads the centre)
    const uint32_t wb = boff + buf * Cf::WSTAGE_BYTES + (t * KS + ks) * Cf::WSTEP_BYTES;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < Cf::NP; ++s)
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.a[mt][s]) : "v"(aoff[mt] + toff + s * Cf::PIECE_BYTES));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < Cf::NP; ++s)
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.b[nt][s]) : "v"(wb + s * Cf::WPIECE_BYTES + nt * 32 * 16));
  };
  auto mma = [&](const Frags& f, auto TODS) {
    HDN_ABL_CONV3X3_4()
    constexpr bool tods = decltype(TODS)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x16 h = tods ? accd[DS ? mt : 0][DS ? nt : 0] : acc[mt][nt];
        f32x16 l = tods ? accdl[DS ? mt : 0][DS ? nt : 0] : accl[mt][nt];
        l = mfma(f.a[mt][1], f.b[nt][0], l);
        h = mfma(f.a[mt][0], f.b[nt][0], h);
        l = mfma(f.a[mt][0], f.b[nt][1], l);
        if (tods) accd[DS ? mt : 0][DS ? nt : 0] = h, accdl[DS ? mt : 0][DS ? nt : 0] = l;
        else acc[mt][nt] = h, accl[mt][nt] = l;
      }
  };

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // Stage t = chunk t / 3, kernel row t % 3 of this launch's K range.  Schedule (producers two stages ahead of the consumers):
  //   W(t): global load at stage t - 4 into wr[t & 1], LDS store at stage t - 2 into ring slot t % 3, read at stage t
  //         (and its first step already at the end of stage t - 1);
  //   A(c): global load at stage (c - 1, row 1), split + LDS store at stage (c, row 1)... into image c & 1, i.e. a chunk is
  //         complete two stages before its first read; a slot / image is rewritten at the earliest one barrier after its last read.
  // The loop is unrolled over two chunks = six stages, so that register sets (t & 1), ring slots (t % 3) and images are static.
  constexpr int NSTEP_ROW[3] = {3 * KS, (DS ? 4 : 3) * KS, 3 * KS};   // steps (tap, k step) per kernel row; the middle row also feeds the downsample branch
  // The two roles run separate loops with the same barrier sequence (so that neither role's registers are live in the other's code).
  // J = position of a stage in the six-stage period, `base` = the period's first chunk.
  if (produce) {
    if (!LAZY) {
      load_a(0);
      load_w(0, P0{});
      load_w(1, P1{});
      store_a(0);
    }
    store_w(P0{}, 0);
    store_w(P1{}, 1);
    if (!LAZY && nchunk > 1) load_a(1);
    if (2 < nstage) load_w(2, P0{});
    if (3 < nstage) load_w(3, P1{});
    __syncthreads();
    auto stage_p = [&](int base, auto Jc) {
      constexpr int J = decltype(Jc)::value, ky = J % 3, ab = (J / 3) & 1;
      const int chunk = base + J / 3, stage = chunk * 3 + ky;
      if (stage + 2 < nstage) store_w(std::integral_constant<int, J & 1>{}, (J + 2) % 3);   // W(stage + 2): that slot was read last in stage - 1
      if (stage + 4 < nstage) load_w(stage + 4, std::integral_constant<int, J & 1>{});
      if (!LAZY && ky == 1 && chunk + 1 < nchunk) {         // (LAZY: both images were filled before the loop)
        store_a(1 - ab);                                   // A(chunk + 1): that image was read last in chunk - 1
        if (chunk + 2 < nchunk) load_a(chunk + 2);
      }
      if (stage + 1 < nstage) __syncthreads();
    };
    int c2 = 0;
    HDN_ABL_CONV3X3_5()
#pragma unroll 1
    for (; c2 + 1 < nchunk; c2 += 2) static_for<6>([&](auto Jc) { stage_p(c2, Jc); });
    if (c2 < nchunk) static_for<3>([&](auto Jc) { stage_p(c2, Jc); });   // an odd chunk count: one more chunk, at position 0 of the period again
  } else {
    __syncthreads();
    // Fragment sets: a step's fragments are read AHEAD steps before its MFMAs (across stage boundaries too: the next stage's
    // images were complete a barrier ago).  The sets rotate with the step number; a period's step count must be a multiple of
    // NSETS for the indices to be static.  One step ahead is what fits: with three sets (two steps ahead) the consumers spill
    // (256 registers per wave at two waves per SIMD) and nothing is gained (measured with round 3's bf16 form: 30.0 / 31.6 vs
    // 28.7 / 30.2 us at 128 / 256 channels).
    constexpr int NSETS = 2, AHEAD = NSETS - 1;
    constexpr int PERIOD_STEPS = 2 * (NSTEP_ROW[0] + NSTEP_ROW[1] + NSTEP_ROW[2]);
    static_assert(PERIOD_STEPS % NSETS == 0, "fragment sets must line up with the period");
    Frags f[NSETS];
    // the step AHEAD steps after (J, st): position in the period (may run into the next period: taken modulo 6 by the caller), step
    auto ahead = [](int J, int st, int d) constexpr {
      for (; d > 0; --d) {
        if (st + 1 < NSTEP_ROW[J % 3]) ++st;
        else { ++J; st = 0; }
      }
      return std::pair<int, int>(J, st);
    };
    read_frags(f[0], 0, 0, 0, 0, 0);   // the very first fragments
    if constexpr (AHEAD > 1) read_frags(f[1], 0, 1 / KS, 1 % KS, 0, 0);
    auto stage_c = [&](int base, auto Jc) {
      constexpr int J = decltype(Jc)::value, ky = J % 3;
      constexpr int NSTEP = NSTEP_ROW[ky];
      // set of this stage's first step: the steps of a period rotate through the sets without a break
      constexpr int fp0 = ((J >= 1 ? NSTEP_ROW[0] : 0) + (J >= 2 ? NSTEP_ROW[1] : 0) + (J >= 3 ? NSTEP_ROW[2] : 0) + (J >= 4 ? NSTEP_ROW[0] : 0) +
                           (J >= 5 ? NSTEP_ROW[1] : 0)) % NSETS;
      const int chunk = base + J / 3, stage = chunk * 3 + ky;
      static_for<NSTEP>([&](auto STc) {
        constexpr int st = decltype(STc)::value, cur = (fp0 + st) % NSETS;
        constexpr std::pair<int, int> tg = ahead(J, st, AHEAD);
        constexpr int Jn = tg.first, stn = tg.second, kyn = Jn % 3, slotn = Jn % 3, abn = ((Jn % 6) / 3) & 1;
        const bool issue = stage + (Jn - J) < nstage;
        if (issue) read_frags(f[(cur + AHEAD) % NSETS], kyn, stn / KS, stn % KS, slotn, abn);
        // the fragments of this step have landed when at most the AHEAD sets read after them are outstanding (LDS operations retire in
        // order; the counter holds 15 at most: waiting for a few reads of the next step as well is harmless)
        constexpr int OUTSTANDING = AHEAD * (MT + NT) * Cf::NP < 15 ? AHEAD * (MT + NT) * Cf::NP : 15;
        if (issue) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(OUTSTANDING) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int s_ = 0; s_ < Cf::NP; ++s_) asm volatile("" : "+v"(f[cur].a[mt][s_]));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int s_ = 0; s_ < Cf::NP; ++s_) asm volatile("" : "+v"(f[cur].b[nt][s_]));
        if constexpr (DS && st / KS == 3) mma(f[cur], std::true_type{});
        else mma(f[cur], std::false_type{});
      });
      if (stage + 1 < nstage) __syncthreads();
    };
    int c2 = 0;
    HDN_ABL_CONV3X3_6()
#pragma unroll 1
    for (; c2 + 1 < nchunk; c2 += 2) static_for<6>([&](auto Jc) { stage_c(c2, Jc); });
    if (c2 < nchunk) static_for<3>([&](auto Jc) { stage_c(c2, Jc); });
  }

  // ---- epilogue.  C/D layout of v_mfma_f32_32x32x16_f16: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  // The tile goes through LDS once ([pixel][BN] fp32) so that the residual is read and the result written as 16 bytes per lane,
  // a pixel's BN channels (contiguous in NHWC) by BN / 4 consecutive lanes.
  __syncthreads();  // every wave is done with the A / W images
  HDN_ABL_CONV3X3_7()
  float* const sO = reinterpret_cast<float*>(smem);
  // (the pixel of accumulator row r is a compile-time constant for each of the two half waves: one multiply-add per store instead of
  //  the mapping's dozen integer operations - round 5)
  if (consume) {
    float* const obase = sO + wm * MT * 32 * Cf::EPI_STRIDE + wn * NT * 32 + li;
    static_for<MT>([&](auto MTc) {
      static_for<16>([&](auto Rc) {
        constexpr int mt = decltype(MTc)::value, r = decltype(Rc)::value, i0 = (r & 3) + 8 * (r >> 2);
        constexpr int row0 = mt * 32 + mrow_to_pixel<Cf>(i0), row1 = mt * 32 + mrow_to_pixel<Cf>(i0 + 4);
        float* const q = obase + (row0 + g * (row1 - row0)) * Cf::EPI_STRIDE;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) q[nt * 32] = join<SD>(acc[mt][nt][r], accl[mt][nt][r]);
      });
    });
  }
  __syncthreads();
  constexpr int ETHREADS = 2 * HDN_BLOCK;   // every thread of the workgroup stores
  const int etid = threadIdx.x;
  constexpr int N4 = BN / 4, TOT4 = BM * N4, EITER = cdiv(TOT4, ETHREADS);
  f4 rv[EITER];
  if (RES) {
#pragma unroll
    for (int q = 0; q < EITER; ++q) {
      const int idx = etid + q * ETHREADS, px = idx / N4, c4 = idx % N4;
      const long long m = min(m0 + px, M - 1);
      rv[q] = *reinterpret_cast<const f4*>(res + m * C + nb * BN + c4 * 4);
    }
  }
#pragma unroll
  for (int q = 0; q < EITER; ++q) {
    const int idx = etid + q * ETHREADS, px = idx / N4, c4 = idx % N4;
    const long long m = m0 + px;
    if (idx < TOT4 && m < M) {
      f4 v = *reinterpret_cast<const f4*>(sO + px * Cf::EPI_STRIDE + c4 * 4);
      if (PARTIAL) {
        *reinterpret_cast<f4*>(out + ((long long)blockIdx.z * M + m) * C + nb * BN + c4 * 4) = v;
      } else {
        v = v + *reinterpret_cast<const f4*>(bias + nb * BN + c4 * 4);
        if (RES) v = v + rv[q];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<f4*>(out + m * C + nb * BN + c4 * 4) = v;
      }
    }
  }
  if (DS) {   // the downsample branch: raw sums (its bias travels with the block's second convolution)
    __syncthreads();
    if (consume) {
      float* const obase = sO + wm * MT * 32 * Cf::EPI_STRIDE + wn * NT * 32 + li;
      static_for<MT>([&](auto MTc) {
        static_for<16>([&](auto Rc) {
          constexpr int mt = decltype(MTc)::value, r = decltype(Rc)::value, i0 = (r & 3) + 8 * (r >> 2);
          constexpr int row0 = mt * 32 + mrow_to_pixel<Cf>(i0), row1 = mt * 32 + mrow_to_pixel<Cf>(i0 + 4);
          float* const q = obase + (row0 + g * (row1 - row0)) * Cf::EPI_STRIDE;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) q[nt * 32] = join<SD>(accd[DS ? mt : 0][DS ? nt : 0][r], accdl[DS ? mt : 0][DS ? nt : 0][r]);
        });
      });
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EITER; ++q) {
      const int idx = etid + q * ETHREADS, px = idx / N4, c4 = idx % N4;
      const long long m = m0 + px;
      if (idx < TOT4 && m < M)
        *reinterpret_cast<f4*>(out2 + ((PARTIAL ? (long long)blockIdx.z * M : 0) + m) * C + nb * BN + c4 * 4) =
            *reinterpret_cast<const f4*>(sO + px * Cf::EPI_STRIDE + c4 * 4);
    }
  }
}

// out = relu(bias + sum over the K slices, in slice order (deterministic) (+ residual)): 16 bytes per lane.  ACT = false: the plain
// sum (the downsample branch).
template <bool RES, bool ACT>
__global__ __launch_bounds__(HDN_BLOCK) void conv3x3_reduce_kernel(const f4* __restrict__ ws, const f4* __restrict__ bias, const f4* __restrict__ res,
                                                                   f4* __restrict__ out, unsigned n4, unsigned c4n, int slices, int res_slices) {
  for (unsigned i = blockIdx.x * HDN_BLOCK + threadIdx.x; i < n4; i += gridDim.x * HDN_BLOCK) {
    // up to 16 slices in flight at once (at B = 1 the launch is a handful of workgroups and nothing but load latency: one round
    // trip instead of one per four slices); the additions stay in slice order, so the sum is the same for any grouping
    f4 v = f4{0.f, 0.f, 0.f, 0.f}, bv = v, rv = v;
    if (ACT) bv = bias[i % c4n];     // (asked for together with the slices, not after them)
    if (ACT && RES) {
      rv = res[i];
      for (int j = 1; j < res_slices; ++j) rv = rv + res[(size_t)j * n4 + i];   // (a downsample branch's raw slices, in slice order)
    }
    for (int z0 = 0; z0 < slices; z0 += 16) {
      f4 a[16];
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (z0 + j < slices) a[j] = ws[(size_t)(z0 + j) * n4 + i];
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (z0 + j < slices) v = (z0 + j == 0) ? a[j] : v + a[j];
    }
    if (ACT) {
      v = v + bv;
      if (RES) v = v + rv;
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    out[i] = v;
  }
}

// K slices for a launch: 1 when the output tiles alone fill the chip, else the smallest power of two (dividing the chunk count)
// that brings the workgroup count to ~one per CU
template <class Cf>
static int k_slices(int B) {
  const long long M = (long long)B * Cf::S * Cf::S;
  const long long tiles = ((M + Cf::BM - 1) / Cf::BM) * Cf::NB;
  static const int target = [] { const char* e = getenv("HDN_CV_SLICE_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 200; }();  // A/B switch
  int z = 1;
  while (tiles * z < target && z * 2 <= Cf::NCHUNK && Cf::NCHUNK % (z * 2) == 0) z *= 2;
  return z;
}

// ... of the chained form: at least NCHUNK / 2, so that a launch's input is at most the two A images (conv3x3_kernel, LAZY)
template <class Cf>
static int k_slices_chain(int B) {
  const int z = k_slices<Cf>(B), zmin = Cf::NCHUNK > 2 ? Cf::NCHUNK / 2 : 1;
  return z > zmin ? z : zmin;
}

template <class Cf>
static size_t workspace_bytes(int B) {
  const int z = k_slices<Cf>(B);
  return z > 1 ? (size_t)(Cf::DS ? 2 : 1) * z * B * Cf::S * Cf::S * Cf::CO * sizeof(float) : 0;
}

template <class Cf, bool SD = false>
static int launch(const float* x, const void* wp, const float* bias, const float* res, float* out, float* out2, float* ws, size_t ws_bytes, int B,
                  hipStream_t stream) {
  const long long M = (long long)B * Cf::S * Cf::S;
  const int z = k_slices<Cf>(B);
  if (z > 1) {  // argument errors before anything touches the device
    if (!ws) return HDN_E_NULL;
    if (ws_bytes < workspace_bytes<Cf>(B) || !aligned16(ws)) return HDN_E_LIMIT;
  }
  static PerDeviceOnce attr;
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    for (const void* fn : {reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 0, false, SD>), reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 1, false, SD>),
                           reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 2, false, SD>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES);
      if (e != hipSuccess) return -(1000 + (int)e);
    }
    attr.set(dev_);
  }
  const dim3 grid((unsigned)((M + Cf::BM - 1) / Cf::BM), Cf::NB, z), blk(2 * HDN_BLOCK);
  const u32x4* w4 = (const u32x4*)wp;
  if (z == 1) {
    if (res) hipLaunchKernelGGL((conv3x3_kernel<Cf, 1, false, SD>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, out, out2, B, Cf::NCHUNK, LazyIn{});
    else hipLaunchKernelGGL((conv3x3_kernel<Cf, 0, false, SD>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, out, out2, B, Cf::NCHUNK, LazyIn{});
    return launch_status();
  }
  float* ws2 = ws + (size_t)z * M * Cf::CO;
  hipLaunchKernelGGL((conv3x3_kernel<Cf, 2, false, SD>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, ws, ws2, B, Cf::NCHUNK / z, LazyIn{});
  const unsigned n4 = (unsigned)(M * Cf::CO / 4);
  const int blocks = (int)((n4 + HDN_BLOCK - 1) / HDN_BLOCK < 1024 ? (n4 + HDN_BLOCK - 1) / HDN_BLOCK : 1024);
  const f4* b4 = (const f4*)bias;
  if (res) hipLaunchKernelGGL((conv3x3_reduce_kernel<true, true>), dim3(blocks), dim3(HDN_BLOCK), 0, stream, (const f4*)ws, b4, (const f4*)res, (f4*)out, n4, Cf::CO / 4, z, 1);
  else hipLaunchKernelGGL((conv3x3_reduce_kernel<false, true>), dim3(blocks), dim3(HDN_BLOCK), 0, stream, (const f4*)ws, b4, (const f4*)res, (f4*)out, n4, Cf::CO / 4, z, 1);
  if (Cf::DS) hipLaunchKernelGGL((conv3x3_reduce_kernel<false, false>), dim3(blocks), dim3(HDN_BLOCK), 0, stream, (const f4*)ws2, b4, (const f4*)nullptr, (f4*)out2, n4, Cf::CO / 4, z, 1);
  return launch_status();
}

// The chained form (small batches): raw K-slice sums out ([z][M][CO], z = k_slices; the downsample branch to out2), whatever z is,
// from an activation (lz.zx == 0) or from the previous convolution's slices (LazyIn).  Nothing is reduced here: the next
// convolution of the chain does that while it stages, the last one's slices go through finish().
template <class Cf, bool SD = false>
static int launch_chain(const float* x, const LazyIn& lz, const void* wp, float* out, float* out2, int B, hipStream_t stream) {
  const long long M = (long long)B * Cf::S * Cf::S;
  const int z = k_slices_chain<Cf>(B);
  if (lz.zx > 0 && Cf::NCHUNK / z > 2) return HDN_E_LIMIT;   // (cannot happen: k_slices_chain; the LAZY staging fills two A images)
  static PerDeviceOnce attr;
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    for (const void* fn : {reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 2, false, SD>), reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 2, true, SD>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES);
      if (e != hipSuccess) return -(1000 + (int)e);
    }
    attr.set(dev_);
  }
  const dim3 grid((unsigned)((M + Cf::BM - 1) / Cf::BM), Cf::NB, z), blk(2 * HDN_BLOCK);
  const u32x4* w4 = (const u32x4*)wp;
  if (lz.zx > 0) hipLaunchKernelGGL((conv3x3_kernel<Cf, 2, true, SD>), grid, blk, Cf::LDS_BYTES, stream, x, w4, nullptr, nullptr, out, out2, B, Cf::NCHUNK / z, lz);
  else hipLaunchKernelGGL((conv3x3_kernel<Cf, 2, false, SD>), grid, blk, Cf::LDS_BYTES, stream, x, w4, nullptr, nullptr, out, out2, B, Cf::NCHUNK / z, LazyIn{});
  return launch_status();
}

static int finish(const float* slices, int z, const float* bias, const float* res, int res_slices, float* out, long long n, int C, hipStream_t stream) {
  const unsigned n4 = (unsigned)(n / 4);
  const int blocks = (int)((n4 + HDN_BLOCK - 1) / HDN_BLOCK < 1024 ? (n4 + HDN_BLOCK - 1) / HDN_BLOCK : 1024);
  if (res) hipLaunchKernelGGL((conv3x3_reduce_kernel<true, true>), dim3(blocks), dim3(HDN_BLOCK), 0, stream, (const f4*)slices, (const f4*)bias, (const f4*)res, (f4*)out, n4, C / 4, z, res_slices);
  else hipLaunchKernelGGL((conv3x3_reduce_kernel<false, true>), dim3(blocks), dim3(HDN_BLOCK), 0, stream, (const f4*)slices, (const f4*)bias, (const f4*)res, (f4*)out, n4, C / 4, z, 1);
  return launch_status();
}


// =====================================================================================================================================
// Round 5: the form for batches that fill the chip (B >= V2_MIN_BATCH, stride 1, C -> C).  What round 4's profile showed for the kernel
// above (profiles/round5_conv3x3.txt): with ONE 32 x 32 MFMA tile per wave and step the LDS moves 8 KB of fragments + the producers'
// weight stores per 6 MFMAs and is the bound (256-channel stage: loop 17.5 us against 6.6 us of matrix pipe), and all of a launch's
// output leaves at its very end.  Here
//   * every consumer wave owns a 64 x 64 output tile (2 x 2 MFMA tiles, 12 MFMAs per step): half the fragment bytes per MFMA;
//   * the four consumers of a workgroup are WM tiles along the pixels x WK slices of K (k step ks of every 16 KS-channel chunk belongs
//     to wave ks % WK), so that a workgroup's tile is 64 WM pixels x 64 channels and the launch still has ~256 workgroups at every stage;
//     the WK partial tiles meet in LDS at the end;
//   * the weights never touch the LDS: a wave's B fragments are its own (no other wave of the workgroup needs that k step, or the
//     other one loads the same lines through the CU's L1), the host packs them in fragment order per (k slice, step), and they stream
//     L2 -> registers two steps ahead (volatile asm loads on a scalar base, explicit vmcnt);
//   * the LDS holds the activations only: image of 16 KS channels, [piece][k step][k half][pixel] x 16 B as above, double-buffered,
//     staged (and split to fp16 pieces) by the four producer waves; ONE barrier per chunk, placed before the MFMAs of a chunk's last
//     step, so that the next chunk's first fragments travel under them;
//   * the producers also own the epilogue: they ask for the residual during the last chunk and, once the partial tiles are in LDS,
//     add them in slice order + bias (+ residual), ReLU, 16-byte NHWC stores.
template <int S_, int C_, int WM_, int WK_, int KS_, int NT_ = 2, int TPW_ = 1>
struct Cfg2 {
  static constexpr int S = S_, C = C_, WM = WM_, WK = WK_, KS = KS_, NT = NT_, TPW = TPW_;
  static_assert(WM * WK == 4 && KS % WK == 0, "four consumer waves: WM pixel tiles x WK k slices");
  // NT = 1 (32 output channels per workgroup, a 64 x 32 tile per wave, 6 MFMAs per step): twice the workgroups for the 4 x 4 stage, whose
  // 1,024 pixels and 512 channels make 128 tiles of 64 x 64 (round 5 first split its K over two workgroups + a reduction launch instead)
  static_assert(NT == 1 || NT == 2, "n tiles per wave");
  static constexpr int BM = 64 * WM, BN = 32 * NT, MT = 2, NP = 2;
  static_assert(C % BN == 0 && C % (16 * KS) == 0, "channel blocking");
  static_assert((BM % S == 0) && ((S * S) % BM == 0 || BM % (S * S) == 0), "a tile is whole rows of one image, or whole images");
  static constexpr int IMGS = BM > S * S ? BM / (S * S) : 1;
  static constexpr int R = BM / (S * IMGS);
  static constexpr int PWV = S + 2, PH = R + 2;
  static constexpr int PW = S == 8 ? 12 : PWV;                       // (bank mapping of the fragment reads: see Cfg)
  static constexpr int IPITCH = PH * PW + (S == 4 ? 4 : 0);
  static constexpr int LPV = IMGS * IPITCH;
  static constexpr int LP = LPV + ((8 / (2 * KS) - LPV % 8) + 8) % 8;
  static constexpr int KG_BYTES = LP * 16, KSTEP_BYTES = 2 * KG_BYTES, PIECE_BYTES = KS * KSTEP_BYTES, A_BYTES = NP * PIECE_BYTES;
  static constexpr int NCHUNK = C / (16 * KS), NB = C / BN;
  static constexpr int SPW = KS / WK, NS = 9 * SPW;                  // k steps / steps (tap, k step) of one wave per chunk
  static constexpr int PER = (NS % 2) ? 2 : 1;                       // chunks per unrolled period (two A register sets)
  // B register sets: a step's fragments travel BSETS - 1 steps ahead.  Half-size steps (NT = 1: 6 MFMAs = 192 clk) need twice the distance.
  static constexpr int BSETS = NT == 2 ? 3 : 6, PF = BSETS - 1;
  static_assert((PER * NS) % BSETS == 0, "the B register sets rotate with the step");
  static constexpr int WSTEP = NT * NP * 64;                         // 16-byte words of one wave step: [n tile][piece][lane]
  static constexpr int WCHUNK = WK * NS * WSTEP;                     // ... of one (channel block, chunk): [k slice][step]
  static constexpr int EPI_STRIDE = BN + 4;
  static constexpr int RED_BYTES = WK * BM * EPI_STRIDE * 4;
  static constexpr int LDS_BYTES = TPW > 1 ? 2 * A_BYTES + RED_BYTES : (2 * A_BYTES > RED_BYTES ? 2 * A_BYTES : RED_BYTES);
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the LDS");
  static_assert(TPW == 1 || (C / (16 * KS)) % 2 == 0, "several tiles per workgroup: an even number of chunks per tile keeps the image parity");
  static constexpr int AITEMS = LP * 2 * KS, AITER = cdiv(AITEMS, HDN_BLOCK);
  static constexpr int E4 = BM * (BN / 4), EITER = cdiv(E4, HDN_BLOCK);   // 16-byte output items per tile / per producer thread
};

// MODE 0: out = relu(conv + bias); 1: out = relu(conv + bias + res); 2: raw sums of the K slice blockIdx.z (`cps` chunks) to out[blockIdx.z][M][C]
template <class Cf, int MODE, bool SD = false>
__global__ __launch_bounds__(2 * HDN_BLOCK) void conv3x3_v2_kernel(const float* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                               const float* __restrict__ res, float* __restrict__ out, int B, int cps) {
  constexpr bool RES = MODE == 1, PARTIAL = MODE == 2;
  constexpr int S = Cf::S, C = Cf::C, BM = Cf::BM, BN = Cf::BN, KS = Cf::KS, WK = Cf::WK, NS = Cf::NS, SPW = Cf::SPW, TPW = Cf::TPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x & (HDN_BLOCK - 1), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool produce = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
  const int li = lane & 31, g = lane >> 5;
  // blockIdx.x = output-channel block (fastest): workgroups are dealt to the 8 XCDs round-robin by their linear id, so an XCD sees
  // NB / 8 (or one) of the channel blocks and its L2 holds that block's weight stream for all the pixel tiles that share it (the 4 x 4
  // stage's 9.4 MB of weights do not fit one 4 MB L2; 1.2 MB per XCD do)
  const int nb = blockIdx.x;
  const long long M = (long long)B * S * S;
  // TPW consecutive pixel tiles per workgroup, one after the other: the producers stage tile t + 1 and store tile t - 1's output while the
  // consumers multiply tile t (a launch's output no longer leaves all at once at its end, and only the first tile's staging is exposed)
  const int tile0 = (int)blockIdx.y * TPW;
  const int ntw = min(TPW, (int)((M + BM - 1) / BM) - tile0);
  const int chunk0 = PARTIAL ? (int)blockIdx.z * cps : 0, nchunk = PARTIAL ? cps : Cf::NCHUNK;
  const int G = ntw * nchunk;                             // chunks of this workgroup, over its tiles
  // the partial tiles meet in LDS: over the dead images when the workgroup has one tile, beside them otherwise
  float* const red = reinterpret_cast<float*>(smem + (TPW > 1 ? 2 * Cf::A_BYTES : 0));

  if (produce) {
    // ------------------------------------------------------------------------------------------------ producers
    f4 av[Cf::AITER][2];
    auto load_a = [&](int gc) {                            // global chunk gc = (tile, chunk)
      const int tile = gc / nchunk, chunk = gc - tile * nchunk;
      const long long m0 = (long long)(tile0 + tile) * BM;
      const int b0 = (int)(m0 / (S * S)), y0 = (int)((m0 % (S * S)) / S);
#pragma unroll
      for (int q = 0; q < Cf::AITER; ++q) {
        const int item = tid + q * HDN_BLOCK;
        const int px = min(item / (2 * KS), Cf::LP - 1), sub = item % (2 * KS);
        const int img = px / Cf::IPITCH, ry = (px % Cf::IPITCH) / Cf::PW, rx = px % Cf::IPITCH % Cf::PW;   // (pad pixels: zeros)
        const int b = b0 + img, y = y0 + ry - 1, xx = rx - 1;
        const bool ok = item < Cf::AITEMS && img < Cf::IMGS && b < B && ry < Cf::PH && y >= 0 && y < S && xx >= 0 && xx < S;
        const f4* src = reinterpret_cast<const f4*>(x + (((size_t)b * S + y) * S + xx) * C + (chunk0 + chunk) * (16 * KS) + sub * 8);
        av[q][0] = ok ? src[0] : f4{0.f, 0.f, 0.f, 0.f};
        av[q][1] = ok ? src[1] : f4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto store_a = [&](int ab) {
#pragma unroll
      for (int q = 0; q < Cf::AITER; ++q) {
        const int item = tid + q * HDN_BLOCK;
        if (item < Cf::AITEMS) {
          const int px = item / (2 * KS), sub = item % (2 * KS);
          unsigned q0[4], q1[4];
          split2x2<SD>(av[q][0].x, av[q][0].y, q0[0], q1[0]);
          split2x2<SD>(av[q][0].z, av[q][0].w, q0[1], q1[1]);
          split2x2<SD>(av[q][1].x, av[q][1].y, q0[2], q1[2]);
          split2x2<SD>(av[q][1].z, av[q][1].w, q0[3], q1[3]);
          unsigned char* dst = smem + ab * Cf::A_BYTES + sub * Cf::KG_BYTES + px * 16;
          *reinterpret_cast<u32x4*>(dst) = u32x4{q0[0], q0[1], q0[2], q0[3]};
          *reinterpret_cast<u32x4*>(dst + Cf::PIECE_BYTES) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        }
      }
    };
    f4 rv[RES ? Cf::EITER : 1];
    auto load_res = [&](int tile) {                        // the residual travels during the tile's last chunk
      if constexpr (RES) {
        const long long m0 = (long long)(tile0 + tile) * BM;
#pragma unroll
        for (int q = 0; q < Cf::EITER; ++q) {
          const int idx = tid + q * HDN_BLOCK, px = idx / (BN / 4), c4 = idx % (BN / 4);
          const long long m = min(m0 + px, M - 1);
          rv[q] = *reinterpret_cast<const f4*>(res + m * C + nb * BN + c4 * 4);
        }
      }
    };
    // sum of the WK partial tiles in slice order (+ bias (+ residual), ReLU), a pixel's BN channels = BN / 4 consecutive lanes
    auto epilogue = [&](int tile) {
      HDN_ABL_CONV3X3_8()
      const long long m0 = (long long)(tile0 + tile) * BM;
#pragma unroll
      for (int q = 0; q < Cf::EITER; ++q) {
        const int idx = tid + q * HDN_BLOCK, px = idx / (BN / 4), c4 = idx % (BN / 4);
        const long long m = m0 + px;
        if (idx < Cf::E4 && m < M) {
          f4 v = *reinterpret_cast<const f4*>(red + px * Cf::EPI_STRIDE + c4 * 4);
#pragma unroll
          for (int w = 1; w < WK; ++w) v = v + *reinterpret_cast<const f4*>(red + (w * BM + px) * Cf::EPI_STRIDE + c4 * 4);
          if (PARTIAL) {
            *reinterpret_cast<f4*>(out + ((long long)blockIdx.z * M + m) * C + nb * BN + c4 * 4) = v;
          } else {
            v = v + *reinterpret_cast<const f4*>(bias + nb * BN + c4 * 4);
            if (RES) v = v + rv[q];
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f4*>(out + m * C + nb * BN + c4 * 4) = v;
          }
        }
      }
    };
    load_a(0);
    store_a(0);
    if (G > 1) load_a(1);
    __syncthreads();                                   // chunk 0 is staged
    for (int gc = 0, tile = 0, c = 0; gc < G; ++gc) {
      if (gc + 1 < G) {
        store_a((gc + 1) & 1);                         // that image was read last in chunk gc - 1, one barrier ago
        if (gc + 2 < G) load_a(gc + 2);
      }
      if (TPW > 1 && c == 0 && tile > 0) epilogue(tile - 1);   // (its partial tiles arrived at the barrier that closed tile - 1; the consumers refill `red`
                                                               //  behind this chunk's barrier at the earliest; rv is free again before load_res below)
      if (c + 1 == nchunk) load_res(tile);
      __syncthreads();                                 // chunk gc + 1 is staged; the consumers have read the last fragment of chunk gc
      if (++c == nchunk) {
        c = 0;
        ++tile;
        __syncthreads();                               // the tile's partial sums are in LDS
      }
    }
    epilogue(ntw - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int wm = wave / WK, wk = wave % WK;
  uint32_t aoff[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int i = (wm * 2 + mt) * 32 + mrow_to_pixel_s1<S>(li);       // pixel inside the workgroup's tile
    const int img = i / (Cf::R * S), yy = (i / S) % Cf::R, xx = i % S;
    aoff[mt] = lds_addr(smem) + g * Cf::KG_BYTES + wk * Cf::KSTEP_BYTES + (img * Cf::IPITCH + yy * Cf::PW + xx) * 16;   // top-left tap, k step wk (+ WK j)
  }
  constexpr int NT = Cf::NT, PF = Cf::PF;
  f32x16 acc[2][NT], accl[2][NT];

  // this wave's weight stream: [channel block][chunk][k slice][step][n tile][piece][lane] x 16 B (the same for every tile of the workgroup)
  const u32x4* const wbase = wp + ((size_t)nb * Cf::NCHUNK + chunk0) * Cf::WCHUNK + (size_t)wk * NS * Cf::WSTEP;
  const uint32_t voff = (uint32_t)lane * 16u;
  u32x4 fb[Cf::BSETS][NT][2], fa[2][2][2];
  // B fragments of (chunk ch, step st) of tile tt; a chunk index past the tile's is the next tile's first chunk — or, behind the last tile,
  // the last step again (which keeps the count of outstanding loads static)
  auto load_b = [&](u32x4 (&b)[NT][2], int tt, int ch, int st) {
    HDN_ABL_CONV3X3_9()
    bool past = false;
    if (ch >= nchunk) {
      if (tt + 1 < ntw) ch -= nchunk;
      else past = true;
    }
    const u32x4* sp = wbase + (size_t)(past ? nchunk - 1 : ch) * Cf::WCHUNK + (size_t)(past ? NS - 1 : st) * Cf::WSTEP;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[0][0]) : "v"(voff), "s"(sp));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[0][1]) : "v"(voff), "s"(sp));
    if constexpr (NT == 2) {
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(b[NT - 1][0]) : "v"(voff), "s"(sp));
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(b[NT - 1][1]) : "v"(voff), "s"(sp));
    }
  };
  // A fragments of step ST (tap ST / SPW, the wave's k step wk + WK (ST % SPW)) of the image at `base` (+ IMG images, when the image
  // parity is static): the whole step-dependent part of the address is the instruction's offset field (no address arithmetic, and
  // nothing the compiler could hoist out of the chunk loop into dozens of live registers)
  auto read_a = [&](u32x4 (&a)[2][2], const uint32_t (&base)[2], auto STc, auto IMGc) {
    constexpr int ST = decltype(STc)::value, IMG = decltype(IMGc)::value, t = ST / SPW;
    constexpr int OFF = ((t / 3) * Cf::PW + (t % 3)) * 16 + WK * (ST % SPW) * Cf::KSTEP_BYTES + IMG * Cf::A_BYTES;
    static_assert(OFF + Cf::PIECE_BYTES < 65536, "ds_read offset field");
    HDN_ABL_CONV3X3_10()
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[mt][0]) : "v"(base[mt]), "n"(OFF));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[mt][1]) : "v"(base[mt]), "n"(OFF + Cf::PIECE_BYTES));
    }
  };
  using I0 = std::integral_constant<int, 0>;
  constexpr bool STATIC_IMG = Cf::PER == 2 && 2 * Cf::A_BYTES < 65536;   // two chunks per period: the image of a step is a constant of the offset field

  static_for<PF>([&](auto Ic) {                         // the first PF steps' fragments
    constexpr int i = decltype(Ic)::value;
    load_b(fb[i], 0, i / NS, i % NS);
  });
  __builtin_amdgcn_s_barrier();                        // chunk 0 is staged
  read_a(fa[0], aoff, I0{}, I0{});
  for (int tt = 0; tt < ntw; ++tt) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = accl[mt][nt][r] = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += Cf::PER) {
      static_for<Cf::PER * NS>([&](auto Pc) {
        constexpr int p = decltype(Pc)::value, cp = p / NS, st = p % NS;
        constexpr int as = p % 2, bs = p % Cf::BSETS;
        const int chunk = c0 + cp;
        uint32_t cur[2] = {aoff[0], aoff[1]}, nxt[2] = {aoff[0], aoff[1]};     // this chunk's image, the next chunk's
        if constexpr (!STATIC_IMG) {
          const uint32_t o = (uint32_t)(chunk & 1) * Cf::A_BYTES;               // (a tile has an even number of chunks when the workgroup has several)
          cur[0] += o; cur[1] += o;
          nxt[0] += o ^ (uint32_t)Cf::A_BYTES; nxt[1] += o ^ (uint32_t)Cf::A_BYTES;
        }
        using CurImg = std::integral_constant<int, STATIC_IMG ? (cp & 1) : 0>;
        using NxtImg = std::integral_constant<int, STATIC_IMG ? ((cp + 1) & 1) : 0>;
        {  // the B fragments PF steps ahead
          constexpr int q = st + PF;
          load_b(fb[(p + PF) % Cf::BSETS], tt, chunk + q / NS, q % NS);
        }
        if constexpr (st + 1 < NS) {
          read_a(fa[as ^ 1], cur, std::integral_constant<int, (st + 1) % NS>{}, CurImg{});
          HDN_ABL_CONV3X3_11(asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(4)" ::"n"(PF * NT * 2) : "memory");)
        } else {
          HDN_ABL_CONV3X3_12(asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PF * NT * 2) : "memory");)
          // the chunk's last fragments are in registers: the producers may overwrite its image, and the next chunk's image is complete
          __builtin_amdgcn_s_barrier();
          if (chunk + 1 < nchunk || tt + 1 < ntw) read_a(fa[as ^ 1], nxt, I0{}, NxtImg{});
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) asm volatile("" : "+v"(fa[as][mt][pc]));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) asm volatile("" : "+v"(fb[bs][nt][pc]));
        // 12 (NT = 1: 6) MFMAs, the three products of an output tile two MFMAs apart
        HDN_ABL_CONV3X3_13()
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) accl[mt][nt] = mfma(fa[as][mt][1], fb[bs][nt][0], accl[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma(fa[as][mt][0], fb[bs][nt][0], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) accl[mt][nt] = mfma(fa[as][mt][0], fb[bs][nt][1], accl[mt][nt]);
      });
    }
    // ---- this wave's partial tile -> LDS.  One tile per workgroup: over the images (every consumer has passed the last chunk's barrier after
    // its last read); several: into their own region, which the producers emptied during this tile's first chunk.
    // C/D layout of v_mfma_f32_32x32x16_f16: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).  The pixel of accumulator row r
    // is a compile-time constant for each of the two half waves: one multiply-add per store, not the mapping's dozen integer operations.
    HDN_ABL_CONV3X3_14_BEGIN
    float* const rbase = red + (wk * BM + wm * 64) * Cf::EPI_STRIDE + li;
    static_for<2>([&](auto MTc) {
      static_for<16>([&](auto Rc) {
        constexpr int mt = decltype(MTc)::value, r = decltype(Rc)::value, i0 = (r & 3) + 8 * (r >> 2);
        constexpr int row0 = mt * 32 + mrow_to_pixel_s1<S>(i0), row1 = mt * 32 + mrow_to_pixel_s1<S>(i0 + 4);
        float* const q = rbase + (row0 + g * (row1 - row0)) * Cf::EPI_STRIDE;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) q[nt * 32] = join<SD>(acc[mt][nt][r], accl[mt][nt][r]);
      });
    });
    HDN_ABL_CONV3X3_14_END
    __syncthreads();                                   // the tile's partial sums are in LDS (the producers take them from there)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the surplus B loads at the tail)
}

template <class Cf>
static int k_slices_v2(int B) {
  const long long M = (long long)B * Cf::S * Cf::S;
  if (Cf::TPW > 1) return 1;       // (several tiles per workgroup: the launch is long enough as it is)
  const long long tiles = ((M + Cf::BM - 1) / Cf::BM) * Cf::NB;
  static const int target = [] { const char* e = getenv("HDN_CV2_SLICE_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 200; }();  // A/B switch
  int z = 1;
  while (tiles * z < target && (Cf::NCHUNK / (z * 2)) % Cf::PER == 0 && Cf::NCHUNK % (z * 2) == 0 && z * 2 <= Cf::NCHUNK) z *= 2;
  return z;
}

template <class Cf, bool SD = false>
static int launch_v2(const float* x, const void* wp, const float* bias, const float* res, float* out, float* ws, size_t ws_bytes, int B, hipStream_t stream) {
  static_assert(Cf::NCHUNK % Cf::PER == 0, "whole periods");
  const long long M = (long long)B * Cf::S * Cf::S;
  const int z = k_slices_v2<Cf>(B);
  if (z > 1) {
    if (!ws) return HDN_E_NULL;
    if (ws_bytes < (size_t)z * M * Cf::C * sizeof(float) || !aligned16(ws)) return HDN_E_LIMIT;
  }
  static PerDeviceOnce attr;
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    for (const void* fn : {reinterpret_cast<const void*>(&conv3x3_v2_kernel<Cf, 0, SD>), reinterpret_cast<const void*>(&conv3x3_v2_kernel<Cf, 1, SD>),
                           reinterpret_cast<const void*>(&conv3x3_v2_kernel<Cf, 2, SD>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES);
      if (e != hipSuccess) return -(1000 + (int)e);
    }
    attr.set(dev_);
  }
  const long long wgs_m = ((M + Cf::BM - 1) / Cf::BM + Cf::TPW - 1) / Cf::TPW;
  if (wgs_m > 65535) return HDN_E_LIMIT;
  const dim3 grid(Cf::NB, (unsigned)wgs_m, z), blk(2 * HDN_BLOCK);
  const u32x4* w4 = (const u32x4*)wp;
  if (z == 1) {
    if (res) hipLaunchKernelGGL((conv3x3_v2_kernel<Cf, 1, SD>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, out, B, Cf::NCHUNK);
    else hipLaunchKernelGGL((conv3x3_v2_kernel<Cf, 0, SD>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, out, B, Cf::NCHUNK);
    return launch_status();
  }
  hipLaunchKernelGGL((conv3x3_v2_kernel<Cf, 2, SD>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, ws, B, Cf::NCHUNK / z);
  return finish(ws, z, bias, res, res ? 1 : 0, out, M * Cf::C, Cf::C, stream);
}

}  // namespace cv
}  // namespace hdn

// Tile configurations.  When the output tiles alone do not fill the chip (the 4 x 4 stage at any batch size, every stage at the
// tracker's B = 1) the K dimension is split over workgroups as well (k_slices) and a second launch reduces the slices.
//                         S   CI   CO  ST  DS    WM WN MT NT KS
using CV_L1  = hdn::cv::Cfg<32, 64, 64, 1, false, 4, 1, 1, 2, 1>;    // 128 pixels (4 rows) x 64 channels: small batches
using CV_L1B = hdn::cv::Cfg<32, 64, 64, 1, false, 4, 1, 2, 2, 1>;    // 256 pixels (8 rows) x 64 channels, 2 x 2 MFMA tiles per wave: B >= 32 (33.9 vs 38.8 us at B = 64); same weight packing
using CV_L2  = hdn::cv::Cfg<16, 128, 128, 1, false, 4, 1, 1, 2, 1>;  // 128 pixels (8 rows) x 64 channels
using CV_L3  = hdn::cv::Cfg<8, 256, 256, 1, false, 2, 2, 1, 1, 2>;   // 64 pixels (one image) x 64 channels
using CV_L4  = hdn::cv::Cfg<4, 512, 512, 1, false, 4, 1, 1, 2, 1>;   // 128 pixels (8 images) x 64 channels, K split 4 ways at B = 64
// first convolution of a stage (stride 2, channels doubled) together with the block's 1x1 / stride-2 downsample branch
using CV_D2  = hdn::cv::Cfg<16, 64, 128, 2, true, 2, 2, 1, 1, 1>;    // 64 output pixels (4 rows of 16) x 64 channels
using CV_D3  = hdn::cv::Cfg<8, 128, 256, 2, true, 2, 2, 1, 1, 1>;    // 64 output pixels (one image) x 64 channels
using CV_D4  = hdn::cv::Cfg<4, 256, 512, 2, true, 2, 2, 1, 1, 1>;    // 64 output pixels (4 images) x 64 channels

// (S = output side, CI input channels, stride) -> configuration
template <class F>
static int cv_dispatch(int S, int CI, int stride, int B, F&& f) {
  if (stride == 1) {
    if (S == 32 && CI == 64) return B >= 32 ? f(CV_L1B{}) : f(CV_L1{});
    if (S == 16 && CI == 128) return f(CV_L2{});
    if (S == 8 && CI == 256) return f(CV_L3{});
    if (S == 4 && CI == 512) return f(CV_L4{});
  } else if (stride == 2) {
    if (S == 16 && CI == 64) return f(CV_D2{});
    if (S == 8 && CI == 128) return f(CV_D3{});
    if (S == 4 && CI == 256) return f(CV_D4{});
  }
  return HDN_E_LIMIT;
}

static_assert(CV_L1::BN == CV_L1B::BN && CV_L1::KS == CV_L1B::KS, "both 64-channel configurations read one weight packing");

extern "C" int hdn_conv3x3_pack_info(int S, int CI, int stride, int* block_n, int* k_steps) {
  return cv_dispatch(S, CI, stride, 1, [&](auto cfg) {
    if (block_n) *block_n = decltype(cfg)::BN;
    if (k_steps) *k_steps = decltype(cfg)::KS;
    return HDN_OK;
  });
}

// bytes of workspace the convolution entry points need for this problem (0: none), or HDN_E_*
extern "C" long long hdn_conv3x3_workspace_bytes(int B, int S, int CI, int stride) {
  if (B <= 0) return HDN_E_SHAPE;
  long long out = 0;
  const int rc = cv_dispatch(S, CI, stride, B, [&](auto cfg) {
    out = (long long)hdn::cv::workspace_bytes<decltype(cfg)>(B);
    return HDN_OK;
  });
  return rc == HDN_OK ? out : rc;
}

static int cv_check(const void* x, const void* w, const void* bias, const void* out, long long n_out) {
  if (!x || !w || !bias || !out) return HDN_E_NULL;
  if (out == x) return HDN_E_ALIAS;
  if (n_out > 0x7fffffffLL) return HDN_E_LIMIT;
  if (!hdn::aligned16(x) || !hdn::aligned16(w) || !hdn::aligned16(out) || !hdn::aligned16(bias)) return HDN_E_LIMIT;
  return HDN_OK;
}

extern "C" int hdn_conv3x3_bias_relu_f32(const float* x, const void* wpacked, const float* bias, const float* residual, float* out, float* workspace,
                                         long long workspace_bytes, int B, int S, int C, int act_domain, void* stream) {
  if (B <= 0 || S <= 0 || C <= 0 || (act_domain != 0 && act_domain != 1)) return HDN_E_SHAPE;
  const int rc = cv_check(x, wpacked, bias, out, (long long)B * S * S * C);  // (out == residual is fine: each element is read before it is written, by the same lane)
  if (rc) return rc;
  if (residual && !hdn::aligned16(residual)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (const int rr = hdn::check_fp16_range(x, (long long)B * S * S * C, s, act_domain)) return rr;
  return cv_dispatch(S, C, 1, B, [&](auto cfg) {
    const size_t wb = workspace_bytes > 0 ? (size_t)workspace_bytes : 0;
    return act_domain ? hdn::cv::launch<decltype(cfg), true>(x, wpacked, bias, residual, out, nullptr, workspace, wb, B, s)
                      : hdn::cv::launch<decltype(cfg), false>(x, wpacked, bias, residual, out, nullptr, workspace, wb, B, s);
  });
}

extern "C" int hdn_conv3x3s2_ds_f32(const float* x, const void* wpacked, const float* bias, float* out, float* out_ds, float* workspace,
                                    long long workspace_bytes, int B, int S, int CI, int act_domain, void* stream) {
  if (B <= 0 || S <= 0 || CI <= 0 || (act_domain != 0 && act_domain != 1)) return HDN_E_SHAPE;
  const int rc = cv_check(x, wpacked, bias, out, (long long)B * S * S * CI * 4);   // (the input has 2S x 2S x CI elements = the output's count x 2)
  if (rc) return rc;
  if (!out_ds) return HDN_E_NULL;
  if (out_ds == out || out_ds == x) return HDN_E_ALIAS;
  if (!hdn::aligned16(out_ds)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (const int rr = hdn::check_fp16_range(x, (long long)B * S * S * CI * 4, s, act_domain)) return rr;
  return cv_dispatch(S, CI, 2, B, [&](auto cfg) {
    const size_t wb = workspace_bytes > 0 ? (size_t)workspace_bytes : 0;
    return act_domain ? hdn::cv::launch<decltype(cfg), true>(x, wpacked, bias, nullptr, out, out_ds, workspace, wb, B, s)
                      : hdn::cv::launch<decltype(cfg), false>(x, wpacked, bias, nullptr, out, out_ds, workspace, wb, B, s);
  });
}


// ---- round 5: the large-batch form (conv3x3_v2_kernel).                  S    C  WM WK KS
using CV2_L1 = hdn::cv::Cfg2<32, 64, 2, 2, 2, 2, 2>;   // 2 tiles of 128 pixels (4 rows) x 64 channels per workgroup, two k slices, chunks of 32 channels
using CV2_L2 = hdn::cv::Cfg2<16, 128, 1, 4, 4, 2, 2>;  // 2 tiles of 64 pixels (4 rows), four k slices, chunks of 64 channels
using CV2_L1S = hdn::cv::Cfg2<32, 64, 4, 1, 2>;        // (HDN_CV2_TPW=1: one tile of 256 pixels (8 rows) per workgroup, chunks of 32 channels)
using CV2_L2S = hdn::cv::Cfg2<16, 128, 2, 2, 4>;       // (HDN_CV2_TPW=1: one tile of 128 pixels (8 rows), two k slices, chunks of 64 channels)
using CV2_L3 = hdn::cv::Cfg2<8, 256, 1, 4, 4>;    // 64 pixels (one image), four k slices
using CV2_L4 = hdn::cv::Cfg2<4, 512, 1, 4, 4, 1>; // 64 pixels (four images) x 32 channels, four k slices: 256 workgroups at B = 64, no K split over workgroups

template <class F>
static int cv2_dispatch(int S, int C, F&& f) {
  static const bool single = [] { const char* e = getenv("HDN_CV2_TPW"); return e && atoi(e) == 1; }();   // A/B switch
  if (S == 32 && C == 64) return single ? f(CV2_L1S{}) : f(CV2_L1{});
  if (S == 16 && C == 128) return single ? f(CV2_L2S{}) : f(CV2_L2{});
  if (S == 8 && C == 256) return f(CV2_L3{});
  if (S == 4 && C == 512) return f(CV2_L4{});
  return HDN_E_LIMIT;
}

extern "C" int hdn_conv3x3_v2_pack_info(int S, int C, int* k_slices, int* k_steps, int* n_tiles) {
  return cv2_dispatch(S, C, [&](auto cfg) {
    if (k_slices) *k_slices = decltype(cfg)::WK;
    if (k_steps) *k_steps = decltype(cfg)::KS;
    if (n_tiles) *n_tiles = decltype(cfg)::NT;
    return HDN_OK;
  });
}

extern "C" long long hdn_conv3x3_v2_workspace_bytes(int B, int S, int C) {
  if (B <= 0) return HDN_E_SHAPE;
  long long out = 0;
  const int rc = cv2_dispatch(S, C, [&](auto cfg) {
    using Cf = decltype(cfg);
    const int z = hdn::cv::k_slices_v2<Cf>(B);
    out = z > 1 ? (long long)z * B * Cf::S * Cf::S * Cf::C * (long long)sizeof(float) : 0;
    return HDN_OK;
  });
  return rc == HDN_OK ? out : rc;
}

extern "C" int hdn_conv3x3_v2_f32(const float* x, const void* wpacked, const float* bias, const float* residual, float* out, float* workspace,
                                  long long workspace_bytes, int B, int S, int C, int act_domain, void* stream) {
  if (B <= 0 || S <= 0 || C <= 0 || (act_domain != 0 && act_domain != 1)) return HDN_E_SHAPE;
  const int rc = cv_check(x, wpacked, bias, out, (long long)B * S * S * C);
  if (rc) return rc;
  if (residual && !hdn::aligned16(residual)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (const int rr = hdn::check_fp16_range(x, (long long)B * S * S * C, s, act_domain)) return rr;
  return cv2_dispatch(S, C, [&](auto cfg) {
    const size_t wb = workspace_bytes > 0 ? (size_t)workspace_bytes : 0;
    return act_domain ? hdn::cv::launch_v2<decltype(cfg), true>(x, wpacked, bias, residual, out, workspace, wb, B, s)
                      : hdn::cv::launch_v2<decltype(cfg), false>(x, wpacked, bias, residual, out, workspace, wb, B, s);
  });
}

// ---- the chained form of the trunk at small batches (hdn_amd.trunk.LazyAct): convolutions hand each other raw K-slice sums
extern "C" int hdn_conv3x3_chain_slices(int B, int S, int CI, int stride) {
  if (B <= 0) return HDN_E_SHAPE;
  return cv_dispatch(S, CI, stride, B, [&](auto cfg) { return hdn::cv::k_slices_chain<decltype(cfg)>(B); });
}

extern "C" int hdn_conv3x3_chain_f32(const float* x, int x_slices, const float* x_bias, const float* x_res, int res_slices, float* x_out, const void* wpacked,
                                     float* out_slices, float* out_ds_slices, int B, int S, int CI, int stride, int act_domain, void* stream) {
  if (B <= 0 || S <= 0 || CI <= 0 || x_slices < 0 || res_slices < 0 || (stride != 1 && stride != 2) || (act_domain != 0 && act_domain != 1)) return HDN_E_SHAPE;
  if (!x || !wpacked || !out_slices || (stride == 2 && !out_ds_slices)) return HDN_E_NULL;
  if (x_slices > 0 && !x_bias) return HDN_E_NULL;
  if (x_slices == 0 && (x_res || x_out)) return HDN_E_SHAPE;          // an activation needs no finishing
  if ((x_res != nullptr) != (res_slices > 0)) return HDN_E_SHAPE;
  if (out_slices == x || out_slices == x_res || out_slices == x_out || (x_out && (x_out == x || x_out == x_res))) return HDN_E_ALIAS;
  const long long n_in = (long long)B * S * S * stride * stride * CI;
  if (n_in > 0x7fffffffLL) return HDN_E_LIMIT;
  for (const void* p : {(const void*)x, wpacked, (const void*)out_slices, (const void*)x_bias, (const void*)x_res, (const void*)x_out, (const void*)out_ds_slices})
    if (p && !hdn::aligned16(p)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // (an activation is checked; an input still in K slices is finished inside the kernel and cannot be seen from here: the guard covers
  //  the chain's first convolution and every un-chained call)
  if (x_slices == 0)
    if (const int rr = hdn::check_fp16_range(x, n_in, s, act_domain)) return rr;
  const hdn::cv::LazyIn lz{x_bias, x_res, x_out, x_slices, res_slices};
  return cv_dispatch(S, CI, stride, B, [&](auto cfg) {
    return act_domain ? hdn::cv::launch_chain<decltype(cfg), true>(x, lz, wpacked, out_slices, out_ds_slices, B, s)
                      : hdn::cv::launch_chain<decltype(cfg), false>(x, lz, wpacked, out_slices, out_ds_slices, B, s);
  });
}

extern "C" int hdn_conv3x3_finish_f32(const float* slices, int n_slices, const float* bias, const float* res, int res_slices, float* out, int B, int S, int C,
                                      void* stream) {
  if (B <= 0 || S <= 0 || C <= 0 || C % 4 || n_slices <= 0 || res_slices < 0) return HDN_E_SHAPE;
  if (!slices || !bias || !out) return HDN_E_NULL;
  if ((res != nullptr) != (res_slices > 0)) return HDN_E_SHAPE;
  if (out == slices) return HDN_E_ALIAS;
  const long long n = (long long)B * S * S * C;
  if (n > 0x7fffffffLL) return HDN_E_LIMIT;
  if (!hdn::aligned16(slices) || !hdn::aligned16(bias) || !hdn::aligned16(out) || (res && !hdn::aligned16(res))) return HDN_E_LIMIT;
  return hdn::cv::finish(slices, n_slices, bias, res, res_slices, out, n, C, static_cast<hipStream_t>(stream));
}
