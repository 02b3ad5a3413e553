// The trunk's three stride-2 stages at large batches (round 5): relu(conv3x3/s2/p1(x, w) + b) AND the block's 1x1/s2 downsample branch from one
// staged input, on the matrix cores, in the producer / consumer form of conv3x3_v2_kernel (conv3x3.hip):
//   x [B, 2S, 2S, C] (channels-last) -> out [B, S, S, 2C] = relu(conv + bias),  out_ds [B, S, S, 2C] = the raw 1x1 / stride-2 product
// Reference: BasicBlock.forward with a downsample branch, homo_estimator/Deep_homography/Oneline_DLTv1/backbone/resnet.py:78-94 (conv1 + bn1 + relu
// and downsample(x); BatchNorm folded into the weights, the branch's bias goes to the block's second convolution: hdn_amd.trunk.FusedBasicBlock).
// hdn_conv3x3s2_ds_f32 (conv3x3.hip, round 4) stays the form for small batches.
//
// fp32 as two fp16 pieces (x = h0 + 2^-11 h1), three piece products into hi / lo accumulators: the error of an fp32 convolution (conv3x3.hip).
// Workgroup = 64 output pixels x 64 output channels: 4 producer waves + 4 consumer waves.  The producers stage the (2R + 1) x (2S + 1) input patch of a
// chunk of 32 input channels as the two pieces' LDS images ([piece][k step][k half][pixel slot] x 16 B, double-buffered) and own the epilogue.  In an
// image row the EVEN padded columns come first, then the odd ones: the 32 pixels of an MFMA tile (consecutive output columns) read consecutive 16-byte
// slots for every tap - (ky, kx) is a constant of the ds_read's offset field: (ky * PW + {0, PWH, 1}[kx]) * 16 - instead of every other one.
// Consumer wave = (pixel half wm, k step wk of the chunk): a 32 x 64 output tile, ten steps per chunk - the nine taps and, with the centre tap's fragment
// again, the downsample branch's weights into a second pair of accumulators.  The weights never touch the LDS: host-packed in fragment order, they travel
// L2 -> registers four steps ahead.  The two k steps' partial tiles meet in LDS (over the dead images) and the producers write both outputs.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "hdn_common.h"
#include "mfma_split.h"

// Measurement hooks (tools/build_variant.sh ... -DHDN_ABLATION -D<experiment>): every site below expands to its production text; the
// experiments' replacement bodies live in ablation/conv3x3s2.inc and are compiled in only under -DHDN_ABLATION, so that editing or adding an
// experiment leaves this translation unit's text (and the hash the committed PMC record carries) unchanged.
#define HDN_ABL_CONV3X3S2_0(...) __VA_ARGS__
#define HDN_ABL_CONV3X3S2_1(...) __VA_ARGS__
#define HDN_ABL_CONV3X3S2_2(...) __VA_ARGS__
#define HDN_ABL_CONV3X3S2_3(...) __VA_ARGS__
#define HDN_ABL_CONV3X3S2_4(...) __VA_ARGS__
#define HDN_ABL_CONV3X3S2_5(...) __VA_ARGS__
#ifdef HDN_ABLATION
#include "ablation/conv3x3s2.inc"
#endif

namespace hdn {
namespace cvs {
using namespace hdn::mc;

template <int SO_, int CI_>
struct CfgS {
  static constexpr int SO = SO_, CI = CI_, CO = 2 * CI_, SI = 2 * SO_;
  static constexpr int BM = 64, BN = 64, NT = 2, KS = 2, WK = 2, WM = 2, NP = 2;
  static_assert(SO == 16 || SO == 8 || SO == 4, "the trunk's three stride-2 stages");
  static constexpr int IMGS = BM > SO * SO ? BM / (SO * SO) : 1;     // images per tile (4 x 4 outputs: four)
  static constexpr int R = BM / (SO * IMGS);                         // output rows of an image in the tile
  static constexpr int PH = 2 * R + 1;                               // padded input rows (one row of padding above, none needed below)
  static constexpr int NE = SO + 1, NO = SO, PWH = NE;               // even / odd padded columns of a row; slot of the first odd one
  // row pitch in slots: 16 lanes of a ds_read_b128 pass = 16 / SO rows of the MFMA tile, 2 PW slots apart - distinct bank groups for PW = 4 mod 8
  // (two rows of 8) and PW = 2 mod 8 (four rows of 4)
  static constexpr int PW = SO == 16 ? 33 : SO == 8 ? 20 : 10;
  static_assert(PW >= NE + NO, "row pitch");
  static constexpr int IPITCH = PH * PW, LPV = IMGS * IPITCH;
  static constexpr int LP = LPV + (4 - LPV % 16 + 16) % 16;          // = 4 mod 16: the four k groups a producer pass writes land on distinct banks
  static constexpr int KG_BYTES = LP * 16, KSTEP_BYTES = 2 * KG_BYTES, PIECE_BYTES = KS * KSTEP_BYTES, A_BYTES = NP * PIECE_BYTES;
  static constexpr int NCHUNK = CI / (16 * KS), NB = CO / BN;
  static constexpr int NS = 10;                                      // steps of a wave per chunk: nine taps + the downsample branch
  static constexpr int BSETS = 5, PF = BSETS - 1;                    // B register sets: a step's fragments travel PF steps ahead
  static_assert(NS % BSETS == 0 && NS % 2 == 0, "register sets rotate with the step");
  static constexpr int WSTEP = NT * NP * 64;                         // 16-byte words of one wave step: [n tile][piece][lane]
  static constexpr int WCHUNK = WK * NS * WSTEP;                     // ... of one (channel block, chunk): [k step][step]
  static constexpr int EPI_STRIDE = BN + 4;
  static constexpr int RED_FLOATS = WK * BM * EPI_STRIDE;            // one output's partial tiles
  static constexpr int RED_BYTES = 2 * RED_FLOATS * 4;
  static constexpr int LDS_BYTES = 2 * A_BYTES > RED_BYTES ? 2 * A_BYTES : RED_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024 && 2 * A_BYTES < 65536 * 2, "LDS");
  static constexpr int AITEMS = LP * 2 * KS, AITER = cdiv(AITEMS, HDN_BLOCK);
  static constexpr int E4 = BM * (BN / 4), EITER = E4 / HDN_BLOCK;
  static_assert(E4 % HDN_BLOCK == 0, "epilogue items");
};

template <class Cf, bool SD = false>
__global__ __launch_bounds__(2 * HDN_BLOCK) void conv3x3s2_v2_kernel(const float* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                                 float* __restrict__ out, float* __restrict__ out_ds, int B) {
  constexpr int SO = Cf::SO, SI = Cf::SI, CI = Cf::CI, CO = Cf::CO, BM = Cf::BM, BN = Cf::BN, KS = Cf::KS, WK = Cf::WK, NS = Cf::NS, NT = Cf::NT, PF = Cf::PF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x & (HDN_BLOCK - 1), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool produce = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
  const int li = lane & 31, g = lane >> 5;
  const int nb = blockIdx.x;                                  // output-channel block, fastest: an XCD keeps its blocks' weight streams in its L2
  const long long M = (long long)B * SO * SO;
  const long long m0 = (long long)blockIdx.y * BM;
  const int b0 = (int)(m0 / (SO * SO)), y0 = (int)((m0 % (SO * SO)) / SO);
  float* const red = reinterpret_cast<float*>(smem);          // the partial tiles meet over the dead images

  if (produce) {
    // ------------------------------------------------------------------------------------------------ producers
    // two register sets: a chunk's pixels are asked for TWO chunks before they are split into the LDS image (a chunk is 10 steps of 6 MFMAs = under a
    // microsecond, less than a round trip to the previous launch's output).  Measured: no change against one set - what a chunk costs here is the
    // producers' own instruction stream (address arithmetic + 32 conversions per item, 6 items per thread: ablations in profiles/round5_conv3x3.txt)
    f4 av[2][Cf::AITER][2];
    // an item = (pixel slot, 8-channel group of the chunk): its source offset, validity and LDS address do not depend on the chunk - worked out once
    // (two divisions by constants per item and chunk otherwise: the producers' instruction stream is what paces a chunk, not the loads' latency)
    uint32_t a_src[Cf::AITER], a_dst[Cf::AITER];
    bool a_ok[Cf::AITER];
#pragma unroll
    for (int q = 0; q < Cf::AITER; ++q) {
      const int item = tid + q * HDN_BLOCK;
      const int px = min(item / (2 * KS), Cf::LPV - 1), sub = item % (2 * KS);
      const int img = px / Cf::IPITCH, ry = (px % Cf::IPITCH) / Cf::PW, sl = px % Cf::IPITCH % Cf::PW;
      const int pc = sl < Cf::NE ? 2 * sl : 2 * (sl - Cf::NE) + 1;                 // padded column of the slot
      const int b = b0 + img, y = 2 * y0 + ry - 1, xx = pc - 1;
      a_ok[q] = item < Cf::AITEMS && item / (2 * KS) < Cf::LPV && sl < Cf::NE + Cf::NO && b < B && y >= 0 && y < SI && xx >= 0 && xx < SI;
      a_src[q] = a_ok[q] ? (uint32_t)(((b * SI + y) * SI + xx) * CI + sub * 8) : 0u;       // (floats; the whole input is < 2^31 of them)
      a_dst[q] = (uint32_t)((item % (2 * KS)) * Cf::KG_BYTES + (item / (2 * KS)) * 16);
    }
    auto load_a = [&](int chunk, auto SETc) {
      constexpr int set = decltype(SETc)::value;
      HDN_ABL_CONV3X3S2_0()
#pragma unroll
      for (int q = 0; q < Cf::AITER; ++q) {
        const f4* src = reinterpret_cast<const f4*>(x + a_src[q] + chunk * (16 * KS));
        av[set][q][0] = a_ok[q] ? src[0] : f4{0.f, 0.f, 0.f, 0.f};
        av[set][q][1] = a_ok[q] ? src[1] : f4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto store_a = [&](int ab, auto SETc) {
      constexpr int set = decltype(SETc)::value;
#pragma unroll
      for (int q = 0; q < Cf::AITER; ++q) {
        if (tid + q * HDN_BLOCK < Cf::AITEMS) {
          unsigned q0[4], q1[4];
          split2x2<SD>(av[set][q][0].x, av[set][q][0].y, q0[0], q1[0]);
          split2x2<SD>(av[set][q][0].z, av[set][q][0].w, q0[1], q1[1]);
          split2x2<SD>(av[set][q][1].x, av[set][q][1].y, q0[2], q1[2]);
          split2x2<SD>(av[set][q][1].z, av[set][q][1].w, q0[3], q1[3]);
          unsigned char* dst = smem + ab * Cf::A_BYTES + a_dst[q];
          *reinterpret_cast<u32x4*>(dst) = u32x4{q0[0], q0[1], q0[2], q0[3]};
          *reinterpret_cast<u32x4*>(dst + Cf::PIECE_BYTES) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        }
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    load_a(0, S0{});
    if (Cf::NCHUNK > 1) load_a(1, S1{});
    store_a(0, S0{});
    if (Cf::NCHUNK > 2) load_a(2, S0{});
    __syncthreads();                                   // chunk 0 is staged
    static_for<Cf::NCHUNK>([&](auto Cc) {              // chunk c + 1 -> its image (read last in chunk c - 1, one barrier ago); chunk c + 3 on its way
      constexpr int c = decltype(Cc)::value;
      using Set = std::integral_constant<int, (c + 1) & 1>;
      if constexpr (c + 1 < Cf::NCHUNK) {
        store_a((c + 1) & 1, Set{});
        if constexpr (c + 3 < Cf::NCHUNK) load_a(c + 3, Set{});
      }
      __syncthreads();                                 // chunk c + 1 is staged; the consumers have read the last fragment of chunk c
    });
    __syncthreads();                                   // both outputs' partial tiles are in LDS
    // sum of the WK partial tiles in k order; the 3 x 3 output + bias, ReLU; the downsample branch raw.  A pixel's 64 channels = 16 consecutive lanes
    HDN_ABL_CONV3X3S2_1()
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const float* const rd = red + o * Cf::RED_FLOATS;
      float* const dst = o ? out_ds : out;
#pragma unroll
      for (int q = 0; q < Cf::EITER; ++q) {
        const int idx = tid + q * HDN_BLOCK, px = idx / (BN / 4), c4 = idx % (BN / 4);
        const long long m = m0 + px;
        if (m < M) {
          f4 v = *reinterpret_cast<const f4*>(rd + px * Cf::EPI_STRIDE + c4 * 4);
#pragma unroll
          for (int w = 1; w < WK; ++w) v = v + *reinterpret_cast<const f4*>(rd + (w * BM + px) * Cf::EPI_STRIDE + c4 * 4);
          if (o == 0) {
            v = v + *reinterpret_cast<const f4*>(bias + nb * BN + c4 * 4);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
          *reinterpret_cast<f4*>(dst + m * CO + nb * BN + c4 * 4) = v;
        }
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int wm = wave / WK, wk = wave % WK;
  uint32_t aoff;
  {
    const int p = wm * 32 + li;                               // pixel inside the workgroup's tile (MFMA row = lane & 31)
    const int img = p / (Cf::R * SO), yy = (p / SO) % Cf::R, xx = p % SO;
    aoff = lds_addr(smem) + g * Cf::KG_BYTES + wk * Cf::KSTEP_BYTES + (img * Cf::IPITCH + 2 * yy * Cf::PW + xx) * 16;   // tap (0, 0): padded (2 yy, 2 xx)
  }
  f32x16 acc[NT], accl[NT], dacc[NT], daccl[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = accl[nt][r] = dacc[nt][r] = daccl[nt][r] = 0.f;

  // this wave's weight stream: [channel block][chunk][k step][step][n tile][piece][lane] x 16 B
  const u32x4* const wbase = wp + (size_t)nb * Cf::NCHUNK * Cf::WCHUNK + (size_t)wk * NS * Cf::WSTEP;
  const uint32_t voff = (uint32_t)lane * 16u;
  u32x4 fb[Cf::BSETS][NT][2], fa[2][2];
  auto load_b = [&](u32x4 (&b)[NT][2], int ch, int st) {    // a step past the last chunk: the last step again (keeps the count of loads in flight static)
    HDN_ABL_CONV3X3S2_2()
    const bool past = ch >= Cf::NCHUNK;
    const u32x4* sp = wbase + (size_t)(past ? Cf::NCHUNK - 1 : ch) * Cf::WCHUNK + (size_t)(past ? NS - 1 : st) * Cf::WSTEP;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[0][0]) : "v"(voff), "s"(sp));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[0][1]) : "v"(voff), "s"(sp));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(b[1][0]) : "v"(voff), "s"(sp));
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(b[1][1]) : "v"(voff), "s"(sp));
  };
  // A fragments of step ST (tap ST; step 9 = the downsample branch = the centre tap's fragment) from the image at `base`
  auto read_a = [&](u32x4 (&a)[2], uint32_t base, auto STc) {
    constexpr int ST = decltype(STc)::value, t = ST == 9 ? 4 : ST, ky = t / 3, kx = t % 3;
    constexpr int OFF = (ky * Cf::PW + (kx == 0 ? 0 : kx == 1 ? Cf::PWH : 1)) * 16;
    static_assert(OFF + Cf::PIECE_BYTES < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[0]) : "v"(base), "n"(OFF));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[1]) : "v"(base), "n"(OFF + Cf::PIECE_BYTES));
  };
  using I0 = std::integral_constant<int, 0>;

  static_for<PF>([&](auto Ic) {                         // the first PF steps' fragments
    constexpr int i = decltype(Ic)::value;
    load_b(fb[i], i / NS, i % NS);
  });
  __builtin_amdgcn_s_barrier();                        // chunk 0 is staged
  read_a(fa[0], aoff, I0{});
  for (int chunk = 0; chunk < Cf::NCHUNK; ++chunk) {
    const uint32_t cur = aoff + (uint32_t)(chunk & 1) * Cf::A_BYTES, nxt = aoff + (uint32_t)((chunk + 1) & 1) * Cf::A_BYTES;
    static_for<NS>([&](auto Pc) {
      constexpr int st = decltype(Pc)::value, as = st % 2, bs = st % Cf::BSETS;
      {  // the B fragments PF steps ahead
        constexpr int q = st + PF;
        load_b(fb[(st + PF) % Cf::BSETS], chunk + q / NS, q % NS);
      }
      if constexpr (st + 1 < NS) {
        read_a(fa[as ^ 1], cur, std::integral_constant<int, st + 1>{});
        HDN_ABL_CONV3X3S2_3(asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(2)" ::"n"(PF * NT * 2) : "memory");)
      } else {
        HDN_ABL_CONV3X3S2_4(asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PF * NT * 2) : "memory");)
        // the chunk's last fragments are in registers: the producers may overwrite its image, and the next chunk's image is complete
        __builtin_amdgcn_s_barrier();
        if (chunk + 1 < Cf::NCHUNK) read_a(fa[as ^ 1], nxt, I0{});
      }
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) asm volatile("" : "+v"(fa[as][pc]));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) asm volatile("" : "+v"(fb[bs][nt][pc]));
      HDN_ABL_CONV3X3S2_5()
      if constexpr (st < 9) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accl[nt] = mfma(fa[as][1], fb[bs][nt][0], accl[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma(fa[as][0], fb[bs][nt][0], acc[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accl[nt] = mfma(fa[as][0], fb[bs][nt][1], accl[nt]);
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) daccl[nt] = mfma(fa[as][1], fb[bs][nt][0], daccl[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dacc[nt] = mfma(fa[as][0], fb[bs][nt][0], dacc[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) daccl[nt] = mfma(fa[as][0], fb[bs][nt][1], daccl[nt]);
      }
    });
  }
  // ---- this wave's two partial tiles -> LDS, over the images (every consumer has passed the last chunk's barrier after its last read).
  // C/D layout of v_mfma_f32_32x32x16_f16: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  float* const rbase = red + (wk * BM + wm * 32 + 4 * g) * Cf::EPI_STRIDE + li;
  static_for<16>([&](auto Rc) {
    constexpr int r = decltype(Rc)::value, row = (r & 3) + 8 * (r >> 2);
    float* const q = rbase + row * Cf::EPI_STRIDE;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      q[nt * 32] = join<SD>(acc[nt][r], accl[nt][r]);
      q[Cf::RED_FLOATS + nt * 32] = join<SD>(dacc[nt][r], daccl[nt][r]);
    }
  });
  __syncthreads();                                     // the partial sums are in LDS (the producers take them from there)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the surplus B loads at the tail)
}

template <class Cf, bool SD = false>
static int launch(const float* x, const void* wp, const float* bias, float* out, float* out_ds, int B, hipStream_t stream) {
  const long long M = (long long)B * Cf::SO * Cf::SO;
  static PerDeviceOnce attr;
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_v2_kernel<Cf, SD>), hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr.set(dev_);
  }
  if ((M + Cf::BM - 1) / Cf::BM > 65535) return HDN_E_LIMIT;      // grid.y (as launch_v2 of conv3x3.hip)
  const dim3 grid(Cf::NB, (unsigned)((M + Cf::BM - 1) / Cf::BM)), blk(2 * HDN_BLOCK);
  hipLaunchKernelGGL((conv3x3s2_v2_kernel<Cf, SD>), grid, blk, Cf::LDS_BYTES, stream, x, static_cast<const u32x4*>(wp), bias, out, out_ds, B);
  return launch_status();
}

}  // namespace cvs
}  // namespace hdn

// wpacked (hdn_amd.trunk.pack_conv3x3s2_ds_v2): [2C / 64][C / 32 chunks][2 k steps][10 steps][2 n tiles][2 pieces][k half g][n][8] fp16; element e of
// lane (g, n) of (block nb, chunk, k step wk, step t, n tile nt) = piece of w[co = 64 nb + 32 nt + n][ci = 32 chunk + 16 wk + 8 g + e][tap t] for
// t < 9 (t = 3 ky + kx), of the downsample branch's w_ds[co][ci] for t = 9.
extern "C" int hdn_conv3x3s2_v2_f32(const float* x, const void* wpacked, const float* bias, float* out, float* out_ds, int B, int S, int CI, int act_domain,
                                    void* stream) {
  if (B <= 0 || S <= 0 || CI <= 0 || (act_domain != 0 && act_domain != 1)) return HDN_E_SHAPE;
  if (!x || !wpacked || !bias || !out || !out_ds) return HDN_E_NULL;
  if (out == x || out_ds == x || out_ds == out) return HDN_E_ALIAS;
  const long long n_in = (long long)B * S * S * CI * 4;       // the input has 2S x 2S x CI elements = an output's count x 2
  if (n_in > 0x7fffffffLL) return HDN_E_LIMIT;
  for (const void* p : {(const void*)x, wpacked, (const void*)bias, (const void*)out, (const void*)out_ds})
    if (!hdn::aligned16(p)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (const int rr = hdn::check_fp16_range(x, n_in, s, act_domain)) return rr;
  auto go = [&](auto cfg) {
    using Cf = decltype(cfg);
    return act_domain ? hdn::cvs::launch<Cf, true>(x, wpacked, bias, out, out_ds, B, s) : hdn::cvs::launch<Cf, false>(x, wpacked, bias, out, out_ds, B, s);
  };
  if (S == 16 && CI == 64) return go(hdn::cvs::CfgS<16, 64>{});
  if (S == 8 && CI == 128) return go(hdn::cvs::CfgS<8, 128>{});
  if (S == 4 && CI == 256) return go(hdn::cvs::CfgS<4, 256>{});
  return HDN_E_LIMIT;
}
