// conv_search of the correlation heads at the tracker's B = 1: a 3x3 / stride 1 / no padding convolution of a 256-channel
// search feature map with the BatchNorm-folded weights of DepthwiseXCorr.conv_search (hdn/models/head/ban.py:55-59,75), + bias + ReLU, written as
// contiguous NCHW planes - the layout the depthwise correlation kernels read.  The cls and loc branches of a level share their input, so the host
// concatenates their weights (CO = 2 x 256) and the three levels of a head are the three problems of ONE launch.
//
// PyTorch / MIOpen run this shape (1 x 256 x 31 x 31 -> 512 x 29 x 29, fp32) as im2col + GEMM or an NHWC implicit GEMM + transposes: 95 / 81 us per
// head for the three levels, + 12 us for the bias / ReLU pass (tools/experiments/exp_head_profile.py).  Here it is an implicit GEMM on the matrix cores
// with fp32 carried as two fp16 pieces (x = h0 + 2^-11 h1, three piece products into hi / lo accumulators; conv3x3.hip has the error analysis: the
// result has the error of an fp32 convolution):
//   D[co][pixel] = sum over (tap, ci) W[co][tap][ci] * X[ci][pixel shifted by the tap]      M = 32 output channels, N = 64 output pixels per workgroup
// * A = weights: split and laid out in fragment order by the host (hdn_amd.heads._pack_conv_search), streamed straight from L2 into registers, a whole
//   64-channel chunk (9 taps x 2 pieces) ahead; volatile asm loads + explicit s_waitcnt (the compiler would sink every load next to its MFMA).
// * B = activations: the input patch of the pixel tile (its output rows + 2, full width), 64 channels at a time, split while it is staged into an
//   LDS image [piece][k step][k half][patch pixel] x 16 B, double-buffered; a tap is a uniform address offset on a per-lane base.
// * Roles: waves 0-3 issue MFMAs (consumers), waves 4-7 stage the next chunk (producers): their vmcnt counters do not meet.  The four consumers split
//   K: consumer w owns the 16-channel slice w of every chunk for all 9 taps and the WHOLE 32 x 64 tile (every fragment is read once per workgroup);
//   their partial sums meet in LDS at the end, where bias and ReLU are applied and 64 consecutive pixels of a channel leave as one coalesced row.
#include <type_traits>
#include <utility>

#include "hdn_common.h"
#include "mfma_split.h"

namespace hdn {
namespace hc {
using namespace hdn::mc;


constexpr int MAX_PROBLEMS = 4, CI = 256, CHUNK = 64, NCHUNK = CI / CHUNK, TILE_M = 32, TILE_N = 64, NTAP = 9;
constexpr int LP_MAX = 224;                                   // patch pixels an LDS image holds
// one 16-byte slot of padding per (k step, k half) sub-image: the 8 sub-images of a pixel then start 16 bytes apart modulo 128, so the
// channels-last staging (8 consecutive lanes = the 8 channel groups of one pixel) writes 8 different 16-byte columns of the bank row
constexpr int KH_BYTES = (LP_MAX + 1) * 16, KSTEP_BYTES = 2 * KH_BYTES, PIECE_BYTES = 4 * KSTEP_BYTES, IMG_BYTES = 2 * PIECE_BYTES;   // [piece][k step][k half][pixel] x 16 B
constexpr int RED_BYTES = 4 * TILE_M * TILE_N * 4;
constexpr int LDS_BYTES = 2 * IMG_BYTES > RED_BYTES ? 2 * IMG_BYTES : RED_BYTES;

struct Ptrs {
  const float* x[MAX_PROBLEMS];
  float* out[MAX_PROBLEMS];
};


// the 18 A fragments (9 taps x 2 pieces) of one (chunk, k slice): contiguous in the packed weights, 1 KB each
__device__ __forceinline__ void load_a(u32x4 (&a)[NTAP][2], const u32x4* wa) {
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[t][pc]) : "v"(wa + (t * 2 + pc) * 64));
}
// fragments of tap T have landed once at most PENDING younger loads are outstanding
template <int PENDING>
__device__ __forceinline__ void wait_a(u32x4& a0, u32x4& a1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a0), "+v"(a1) : "n"(PENDING));
}

__global__ __launch_bounds__(512) void head_conv_kernel(Ptrs P, const u32x4* __restrict__ wp, const float* __restrict__ bias, int CO, int Hi, int Wi,
                                                        long long sc, long long sy, long long sx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) & 3));
  const bool produce = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
  const int li = lane & 31, g = lane >> 5;
  const int Ho = Hi - 2, Wo = Wi - 2, Pn = Ho * Wo;
  const int p0 = blockIdx.x * TILE_N, cb = blockIdx.y, prob = blockIdx.z;
  const float* __restrict__ x = P.x[prob];
  const int y0 = p0 / Wo;                                     // first output row of the tile = first input row of its patch
  const int y1 = min(p0 + TILE_N - 1, Pn - 1) / Wo;
  const int LP = (y1 - y0 + 3) * Wi;                          // patch pixels (host checked: <= LP_MAX)

  if (produce) {
    // ---- producers: stage chunk c of the patch (64 channels, split to fp16 pieces) into image c & 1.
    // item = (patch pixel, group of 8 channels), 8 LP items per chunk.  The lane -> item order follows the memory layout so that a load instruction
    // of a wave covers contiguous bytes: NCHW (a channel's patch = LP contiguous floats): pixel fastest, 8 scalar loads at the channel stride;
    // channels-last (a pixel's 64 chunk channels = 256 contiguous bytes): channel group fastest, two 16-byte loads.
    constexpr int ITER = (8 * LP_MAX + 255) / 256;
    const bool cl = sc == 1;                                  // (uniform)
    long long goff[ITER];
    int loff[ITER];
#pragma unroll
    for (int q = 0; q < ITER; ++q) {
      const int item = tid + q * 256;
      const int pxr = cl ? item >> 3 : item % LP, cg = cl ? item & 7 : min(item / LP, 7);
      const int px = min(pxr, LP - 1);
      const int yi = y0 + px / Wi, xi = px % Wi;
      goff[q] = (long long)(cg * 8) * sc + (long long)yi * sy + (long long)xi * sx;
      loff[q] = (item < 8 * LP) ? (cg >> 1) * KSTEP_BYTES + (cg & 1) * KH_BYTES + px * 16 : -1;
    }
    // chunk c + 1 travels (in the other register set) while chunk c is split, stored and waited for: one exposed round trip per workgroup
    float v[2][ITER][8];
    auto load_chunk = [&](int c, auto SET) {
      constexpr int set = decltype(SET)::value;
      if (cl) {
#pragma unroll
        for (int q = 0; q < ITER; ++q) {
          const f4* src = reinterpret_cast<const f4*>(x + goff[q] + c * CHUNK);     // 32-byte aligned: 256 channels, groups of 8
          const f4 lo4 = src[0], hi4 = src[1];
          v[set][q][0] = lo4.x; v[set][q][1] = lo4.y; v[set][q][2] = lo4.z; v[set][q][3] = lo4.w;
          v[set][q][4] = hi4.x; v[set][q][5] = hi4.y; v[set][q][6] = hi4.z; v[set][q][7] = hi4.w;
        }
      } else {
#pragma unroll
        for (int q = 0; q < ITER; ++q)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[set][q][j] = x[goff[q] + (long long)(c * CHUNK + j) * sc];
      }
    };
    auto store_chunk = [&](int c, auto SET) {
      constexpr int set = decltype(SET)::value;
      unsigned char* img = smem + (c & 1) * IMG_BYTES;
#pragma unroll
      for (int q = 0; q < ITER; ++q) {
        unsigned q0[4], q1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2x2(v[set][q][2 * j], v[set][q][2 * j + 1], q0[j], q1[j]);
        if (loff[q] >= 0) {
          *reinterpret_cast<u32x4*>(img + loff[q]) = u32x4{q0[0], q0[1], q0[2], q0[3]};
          *reinterpret_cast<u32x4*>(img + loff[q] + PIECE_BYTES) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        }
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    static_assert(NCHUNK == 4, "the chunk loop is written out");
    load_chunk(0, S0{});
    load_chunk(1, S1{});
    store_chunk(0, S0{});
    __syncthreads();                                          // chunk 0 is complete
    load_chunk(2, S0{});
    store_chunk(1, S1{});
    __syncthreads();                                          // chunk 1 is complete, the consumers are done with chunk 0
    load_chunk(3, S1{});
    store_chunk(2, S0{});
    __syncthreads();
    store_chunk(3, S1{});
    __syncthreads();
    __syncthreads();                                          // (the consumers' last barrier: they are done with the images)
  } else {
    // ---- consumers: K slice `wave` of every chunk, all 9 taps, the whole 32 x 64 tile
    f32x16 hi[2], lo[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) hi[nt][r] = lo[nt][r] = 0.f;
    // per-lane patch pixel of output pixel p0 + 32 nt + li (clamped to the last pixel: computed, not stored)
    uint32_t bbase[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int p = min(p0 + nt * 32 + li, Pn - 1), yo = p / Wo, xo = p - yo * Wo;
      bbase[nt] = (uint32_t)(((yo - y0) * Wi + xo) * 16 + wave * KSTEP_BYTES + g * KH_BYTES);
    }
    // packed weights [problem][channel block][chunk][k slice][tap][piece][lane]
    const u32x4* wa = wp + ((((size_t)prob * (CO / TILE_M) + cb) * NCHUNK) * 4 + wave) * (NTAP * 2 * 64) + lane;
    constexpr size_t CHUNK_WORDS = (size_t)4 * NTAP * 2 * 64;
    u32x4 a[2][NTAP][2];
    load_a(a[0], wa);
    auto chunk = [&](int c, auto BUF, auto NEXT) {
      constexpr int bsel = decltype(BUF)::value;
      constexpr bool next = decltype(NEXT)::value;           // the following chunk's 18 loads are in flight behind this chunk's
      const unsigned char* img = smem + (c & 1) * IMG_BYTES;
      static_for<NTAP>([&](auto Tc) {
        constexpr int t = decltype(Tc)::value;
        const int toff = ((t / 3) * Wi + (t % 3)) * 16;
        u32x4 b[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          b[nt][0] = *reinterpret_cast<const u32x4*>(img + bbase[nt] + toff);
          b[nt][1] = *reinterpret_cast<const u32x4*>(img + bbase[nt] + toff + PIECE_BYTES);
        }
        wait_a<(NTAP - 1 - t) * 2 + (next ? NTAP * 2 : 0)>(a[bsel][t][0], a[bsel][t][1]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          lo[nt] = mfma(a[bsel][t][1], b[nt][0], lo[nt]);
          hi[nt] = mfma(a[bsel][t][0], b[nt][0], hi[nt]);
          lo[nt] = mfma(a[bsel][t][0], b[nt][1], lo[nt]);
        }
      });
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    static_assert(NCHUNK == 4, "the chunk loop is written out");
    __syncthreads();                                          // chunk 0 staged
    load_a(a[1], wa + CHUNK_WORDS);
    chunk(0, B0{}, std::true_type{});
    __syncthreads();                                          // chunk 1 staged, chunk 0's image free
    load_a(a[0], wa + 2 * CHUNK_WORDS);
    chunk(1, B1{}, std::true_type{});
    __syncthreads();
    load_a(a[1], wa + 3 * CHUNK_WORDS);
    chunk(2, B0{}, std::true_type{});
    __syncthreads();
    chunk(3, B1{}, std::false_type{});
    __syncthreads();                                          // every consumer is done with the images: the LDS becomes the reduction buffer
    float* red = reinterpret_cast<float*>(smem) + wave * (TILE_M * TILE_N);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r)                            // C/D layout: row (r & 3) + 8 (r >> 2) + 4 g = channel, column li = pixel
        red[((r & 3) + 8 * (r >> 2) + 4 * g) * TILE_N + nt * 32 + li] = join(hi[nt][r], lo[nt][r]);
  }
  __syncthreads();
  // ---- all 512 threads: sum of the four K slices in slice order + bias + ReLU -> NCHW rows
  {
    const float* red = reinterpret_cast<const float*>(smem);
    float* __restrict__ out = P.out[prob];
    const int t512 = threadIdx.x;
#pragma unroll
    for (int q = 0; q < (TILE_M * TILE_N) / 512; ++q) {
      const int e = t512 + q * 512, ch = e / TILE_N, px = e % TILE_N;
      float v = red[e];
#pragma unroll
      for (int w = 1; w < 4; ++w) v += red[w * (TILE_M * TILE_N) + e];
      v = fmaxf(v + bias[(size_t)prob * CO + cb * TILE_M + ch], 0.f);
      if (p0 + px < Pn) out[(size_t)(cb * TILE_M + ch) * Pn + p0 + px] = v;
    }
  }
}

}  // namespace hc
}  // namespace hdn

extern "C" int hdn_head_conv3x3_f32(const float* const* xs, const void* w_packed, const float* bias, float* const* outs, int n, int CO, int Hi, int Wi,
                                    int nhwc, void* stream) {
  if (!xs || !w_packed || !bias || !outs) return HDN_E_NULL;
  if (n <= 0 || CO <= 0 || Hi < 3 || Wi < 3) return HDN_E_SHAPE;
  if (n > hdn::hc::MAX_PROBLEMS || CO % hdn::hc::TILE_M != 0 || Hi > 1024 || Wi > 1024) return HDN_E_LIMIT;
  const int Ho = Hi - 2, Wo = Wi - 2;
  // the patch of 64 consecutive output pixels: its output rows + 2, full width
  const int rows = (hdn::hc::TILE_N - 1 + Wo - 1) / Wo + 1 + 2;
  if ((rows < Hi ? rows : Hi) * Wi > hdn::hc::LP_MAX) return HDN_E_LIMIT;
  if (!hdn::aligned16(w_packed)) return HDN_E_LIMIT;
  hdn::hc::Ptrs P{};
  for (int i = 0; i < n; ++i) {
    if (!xs[i] || !outs[i]) return HDN_E_NULL;
    if (static_cast<const void*>(xs[i]) == static_cast<const void*>(outs[i])) return HDN_E_ALIAS;
    P.x[i] = xs[i];
    P.out[i] = outs[i];
    if (const int rr = hdn::check_fp16_range(xs[i], (long long)hdn::hc::CI * Hi * Wi, static_cast<hipStream_t>(stream))) return rr;
  }
  static hdn::PerDeviceOnce attr;
  const int dev_ = hdn::PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hdn::hc::head_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, hdn::hc::LDS_BYTES);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr.set(dev_);
  }
  const long long sc = nhwc ? 1 : (long long)Hi * Wi, sy = nhwc ? (long long)Wi * hdn::hc::CI : Wi, sx = nhwc ? hdn::hc::CI : 1;
  const dim3 grid((Ho * Wo + hdn::hc::TILE_N - 1) / hdn::hc::TILE_N, CO / hdn::hc::TILE_M, n);
  hipLaunchKernelGGL(hdn::hc::head_conv_kernel, grid, dim3(512), hdn::hc::LDS_BYTES, static_cast<hipStream_t>(stream), P, static_cast<const hdn::hc::u32x4*>(w_packed),
                     bias, CO, Hi, Wi, sc, sy, sx);
  return hdn::launch_status();
}
