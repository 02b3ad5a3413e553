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
// (rms 6.6e-8 of the output scale on K = 2,304 sums, fp32 sgemm: 6.6e-8, the bf16 form: 2.8e-8).  Range: |x| < 65,504 (fp16);
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

namespace hdn {
namespace cv {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N - 1
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two fp32 values -> their two fp16 pieces (round-to-nearest-even by v_cvt_pk_f16_f32), packed (lo = first value):
// x = p0 + 2^-11 p1 up to 2^-23 |x|; the residual x - p0 is exact in fp32, and so is its product with 2^11 (6 VALU ops per pair)
constexpr float LO_SCALE = 2048.f, LO_UNSCALE = 1.f / 2048.f;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2x2(float x, float y, unsigned& p0, unsigned& p1) {
  const f2 v = {x, y};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  p0 = __builtin_bit_cast(unsigned, h);
  const f2 r = (v - __builtin_convertvector(h, f2)) * LO_SCALE;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

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
template <class Cf>
__device__ __forceinline__ int mrow_to_pixel(int i) {
  if constexpr (Cf::STRIDE != 1 || Cf::S >= 32) {
    return i;
  } else {
    const int q = i >> 2, grp = (0x96 >> q) & 1, k = ((q >> 1) << 2) | (i & 3);
    if constexpr (Cf::S == 16) return grp * 16 + k;
    else if constexpr (Cf::S == 8) return (2 * (k >> 3) + grp) * 8 + (k & 7);
    else return (k >> 3) * 16 + (2 * ((k >> 2) & 1) + grp) * 4 + (k & 3);
  }
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
template <class Cf, int MODE, bool LAZY = false>
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
#if defined(HDN_ABLATION) && defined(CV_EXP_NOWLOAD)   // measurement build only: tools/build_variant.sh -DHDN_ABLATION -DCV_EXP_NOWLOAD
    if (stage > 3) return;
#endif
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
#if defined(HDN_ABLATION) && defined(CV_EXP_NOALOAD)   // measurement build only: tools/build_variant.sh -DHDN_ABLATION -DCV_EXP_NOALOAD
    if (chunk > 0) return;
#endif
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
#if defined(HDN_ABLATION) && defined(CV_EXP_NOSTAGE)   // measurement build only: tools/build_variant.sh -DHDN_ABLATION -DCV_EXP_NOSTAGE
    return;
#endif
#pragma unroll
    for (int q = 0; q < Cf::AITER; ++q) {
      const int item = tid + q * HDN_BLOCK;
      if (item < Cf::AITEMS) {
        const int px = item / (2 * KS), sub = item % (2 * KS);   // sub = k step * 2 + k half
        unsigned q0[4], q1[4];
        split2x2(av[q][0].x, av[q][0].y, q0[0], q1[0]);
        split2x2(av[q][0].z, av[q][0].w, q0[1], q1[1]);
        split2x2(av[q][1].x, av[q][1].y, q0[2], q1[2]);
        split2x2(av[q][1].z, av[q][1].w, q0[3], q1[3]);
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
      split2x2(s0.x, s0.y, q0[0], q1[0]);
      split2x2(s0.z, s0.w, q0[1], q1[1]);
      split2x2(s1.x, s1.y, q0[2], q1[2]);
      split2x2(s1.z, s1.w, q0[3], q1[3]);
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
#if defined(HDN_ABLATION) && defined(CV_EXP_NOREAD)   // measurement build only: tools/build_variant.sh -DHDN_ABLATION -DCV_EXP_NOREAD
    if (ky + t + ks + buf >= 0) return;
#endif
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
#if defined(HDN_ABLATION) && defined(CV_EXP_NOMFMA)   // measurement build only: tools/build_variant.sh -DHDN_ABLATION -DCV_EXP_NOMFMA
    return;
#endif
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
#pragma unroll 1
    for (; c2 + 1 < nchunk; c2 += 2) static_for<6>([&](auto Jc) { stage_c(c2, Jc); });
    if (c2 < nchunk) static_for<3>([&](auto Jc) { stage_c(c2, Jc); });
  }

  // ---- epilogue.  C/D layout of v_mfma_f32_32x32x16_f16: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  // The tile goes through LDS once ([pixel][BN] fp32) so that the residual is read and the result written as 16 bytes per lane,
  // a pixel's BN channels (contiguous in NHWC) by BN / 4 consecutive lanes.
  __syncthreads();  // every wave is done with the A / W images
  float* const sO = reinterpret_cast<float*>(smem);
  if (consume) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * MT + mt) * 32 + mrow_to_pixel<Cf>((r & 3) + 8 * (r >> 2) + 4 * g);
          sO[row * Cf::EPI_STRIDE + (wn * NT + nt) * 32 + li] = acc[mt][nt][r] + accl[mt][nt][r] * LO_UNSCALE;
        }
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
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (wm * MT + mt) * 32 + mrow_to_pixel<Cf>((r & 3) + 8 * (r >> 2) + 4 * g);
            sO[row * Cf::EPI_STRIDE + (wn * NT + nt) * 32 + li] = accd[DS ? mt : 0][DS ? nt : 0][r] + accdl[DS ? mt : 0][DS ? nt : 0][r] * LO_UNSCALE;
          }
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

template <class Cf>
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
    for (const void* fn : {reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 0>), reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 1>),
                           reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 2>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES);
      if (e != hipSuccess) return -(1000 + (int)e);
    }
    attr.set(dev_);
  }
  const dim3 grid((unsigned)((M + Cf::BM - 1) / Cf::BM), Cf::NB, z), blk(2 * HDN_BLOCK);
  const u32x4* w4 = (const u32x4*)wp;
  if (z == 1) {
    if (res) hipLaunchKernelGGL((conv3x3_kernel<Cf, 1>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, out, out2, B, Cf::NCHUNK, LazyIn{});
    else hipLaunchKernelGGL((conv3x3_kernel<Cf, 0>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, out, out2, B, Cf::NCHUNK, LazyIn{});
    return launch_status();
  }
  float* ws2 = ws + (size_t)z * M * Cf::CO;
  hipLaunchKernelGGL((conv3x3_kernel<Cf, 2>), grid, blk, Cf::LDS_BYTES, stream, x, w4, bias, res, ws, ws2, B, Cf::NCHUNK / z, LazyIn{});
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
template <class Cf>
static int launch_chain(const float* x, const LazyIn& lz, const void* wp, float* out, float* out2, int B, hipStream_t stream) {
  const long long M = (long long)B * Cf::S * Cf::S;
  const int z = k_slices_chain<Cf>(B);
  if (lz.zx > 0 && Cf::NCHUNK / z > 2) return HDN_E_LIMIT;   // (cannot happen: k_slices_chain; the LAZY staging fills two A images)
  static PerDeviceOnce attr;
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    for (const void* fn : {reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 2, false>), reinterpret_cast<const void*>(&conv3x3_kernel<Cf, 2, true>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES);
      if (e != hipSuccess) return -(1000 + (int)e);
    }
    attr.set(dev_);
  }
  const dim3 grid((unsigned)((M + Cf::BM - 1) / Cf::BM), Cf::NB, z), blk(2 * HDN_BLOCK);
  const u32x4* w4 = (const u32x4*)wp;
  if (lz.zx > 0) hipLaunchKernelGGL((conv3x3_kernel<Cf, 2, true>), grid, blk, Cf::LDS_BYTES, stream, x, w4, nullptr, nullptr, out, out2, B, Cf::NCHUNK / z, lz);
  else hipLaunchKernelGGL((conv3x3_kernel<Cf, 2, false>), grid, blk, Cf::LDS_BYTES, stream, x, w4, nullptr, nullptr, out, out2, B, Cf::NCHUNK / z, LazyIn{});
  return launch_status();
}

static int finish(const float* slices, int z, const float* bias, const float* res, int res_slices, float* out, long long n, int C, hipStream_t stream) {
  const unsigned n4 = (unsigned)(n / 4);
  const int blocks = (int)((n4 + HDN_BLOCK - 1) / HDN_BLOCK < 1024 ? (n4 + HDN_BLOCK - 1) / HDN_BLOCK : 1024);
  if (res) hipLaunchKernelGGL((conv3x3_reduce_kernel<true, true>), dim3(blocks), dim3(HDN_BLOCK), 0, stream, (const f4*)slices, (const f4*)bias, (const f4*)res, (f4*)out, n4, C / 4, z, res_slices);
  else hipLaunchKernelGGL((conv3x3_reduce_kernel<false, true>), dim3(blocks), dim3(HDN_BLOCK), 0, stream, (const f4*)slices, (const f4*)bias, (const f4*)res, (f4*)out, n4, C / 4, z, 1);
  return launch_status();
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
                                         long long workspace_bytes, int B, int S, int C, void* stream) {
  if (B <= 0 || S <= 0 || C <= 0) return HDN_E_SHAPE;
  const int rc = cv_check(x, wpacked, bias, out, (long long)B * S * S * C);  // (out == residual is fine: each element is read before it is written, by the same lane)
  if (rc) return rc;
  if (residual && !hdn::aligned16(residual)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return cv_dispatch(S, C, 1, B, [&](auto cfg) {
    return hdn::cv::launch<decltype(cfg)>(x, wpacked, bias, residual, out, nullptr, workspace, workspace_bytes > 0 ? (size_t)workspace_bytes : 0, B, s);
  });
}

extern "C" int hdn_conv3x3s2_ds_f32(const float* x, const void* wpacked, const float* bias, float* out, float* out_ds, float* workspace,
                                    long long workspace_bytes, int B, int S, int CI, void* stream) {
  if (B <= 0 || S <= 0 || CI <= 0) return HDN_E_SHAPE;
  const int rc = cv_check(x, wpacked, bias, out, (long long)B * S * S * CI * 4);   // (the input has 2S x 2S x CI elements = the output's count x 2)
  if (rc) return rc;
  if (!out_ds) return HDN_E_NULL;
  if (out_ds == out || out_ds == x) return HDN_E_ALIAS;
  if (!hdn::aligned16(out_ds)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return cv_dispatch(S, CI, 2, B, [&](auto cfg) {
    return hdn::cv::launch<decltype(cfg)>(x, wpacked, bias, nullptr, out, out_ds, workspace, workspace_bytes > 0 ? (size_t)workspace_bytes : 0, B, s);
  });
}

// ---- the chained form of the trunk at small batches (hdn_amd.trunk.LazyAct): convolutions hand each other raw K-slice sums
extern "C" int hdn_conv3x3_chain_slices(int B, int S, int CI, int stride) {
  if (B <= 0) return HDN_E_SHAPE;
  return cv_dispatch(S, CI, stride, B, [&](auto cfg) { return hdn::cv::k_slices_chain<decltype(cfg)>(B); });
}

extern "C" int hdn_conv3x3_chain_f32(const float* x, int x_slices, const float* x_bias, const float* x_res, int res_slices, float* x_out, const void* wpacked,
                                     float* out_slices, float* out_ds_slices, int B, int S, int CI, int stride, void* stream) {
  if (B <= 0 || S <= 0 || CI <= 0 || x_slices < 0 || res_slices < 0 || (stride != 1 && stride != 2)) return HDN_E_SHAPE;
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
  const hdn::cv::LazyIn lz{x_bias, x_res, x_out, x_slices, res_slices};
  return cv_dispatch(S, CI, stride, B, [&](auto cfg) {
    return hdn::cv::launch_chain<decltype(cfg)>(x, lz, wpacked, out_slices, out_ds_slices, B, s);
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
