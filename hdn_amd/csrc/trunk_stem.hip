// First stage of the homography regressor's ResNet-34 trunk, fused:
//   y = maxpool3x3/s2/p1( relu( conv7x7/s2/p3(x, w) + b ) ),   x [B,2,H,W] -> y [B,64,Hp,Wp]   (127 -> 64 -> 32)
// with eval-mode BatchNorm folded into (w, b) by the host (hdn_amd/trunk.py).  Reference: HomoResNet.forward,
// homo_estimator/Deep_homography/Oneline_DLTv1/backbone/resnet.py:137-194 (conv1 / bn1 / relu / maxpool, :141-147,:183-186).
// MIOpen runs this stage as three kernels with the 64-channel 64x64 conv output (67 MB at B = 64) going through HBM twice;
// here it never leaves the registers.
//
// A WAVE owns (image, block of 16 output channels, strip of pooled rows); lane = conv column (Wc <= 64).  Per pooled row p
// it computes conv rows 2p and 2p+1 (row 2p-1 is kept from the previous iteration) from the 9 input rows they share: an
// input row's 7 taps per lane are loaded once and feed both conv rows; the 16 weights of a tap are one s_load_dwordx16,
// used as SGPR pair operands of 8 v_pk_fma_f32 (two output channels per instruction, the input value broadcast by op_sel).
// Pooling: vertical v_max3 over the three conv rows, horizontal v_max3 with the two neighbour lanes (DPP wave shifts; lanes
// outside the row hold 0, which never wins after the ReLU and every window has a real element), even lanes store.
// Zero padding is selected (v_cndmask), never multiplied in: non-finite pixels stay where the reference has them.
// fp32 throughout; the summation order is (input channel, input row, kx) per output - not MIOpen's, same 1e-5 relative class.
#include <cstdlib>
#include <type_traits>

#include "hdn_common.h"

namespace hdn {
namespace stem {
constexpr int CO = 64, CB = 16, KS = 7, NPAIR = CB / 2;
typedef const float __attribute__((address_space(4))) cfloat;
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const f16v __attribute__((address_space(4), aligned(4))) c16;
typedef float f4e __attribute__((ext_vector_type(4)));

__device__ __forceinline__ const cfloat* opaque(const float* p) {
  uint64_t a = reinterpret_cast<uint64_t>(p);
  asm volatile("" : "+s"(a));
  return (const cfloat*)a;
}
__device__ __forceinline__ float from_prev_lane(float v) {  // lane l <- lane l-1, lane 0 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_next_lane(float v) {  // lane l <- lane l+1, lane 63 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// The weight stream: scalar loads the compiler does not track, so that tap k+1 is in flight while tap k's 72 FMAs run (a
// tracked s_load is waited for with lgkmcnt(0) right where it is issued: SMEM returns out of order).  The FMAs are volatile as
// well: program order = schedule.
template <int OFF>
__device__ __forceinline__ f16v sload16(const cfloat* p) {
  f16v r;
  asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(r) : "s"(p), "n"(OFF));
  return r;
}
__device__ __forceinline__ void swait(f16v& w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w)); }
#define STEM_PAIR(w, k) __builtin_shufflevector(w, w, 2 * (k), 2 * (k) + 1)
// acc[k] += (v, v) * w[k] for the 8 channel pairs of a tap; v = half HALF of the register pair xv
template <int HALF>
__device__ __forceinline__ void fma8(float2v (&acc)[NPAIR], float2v xv, f16v w) {
  if constexpr (HALF == 0)
    asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %8, %10, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %2, %8, %11, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %8, %12, %3 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %4, %8, %13, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %5, %8, %14, %5 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %6, %8, %15, %6 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %7, %8, %16, %7 op_sel_hi:[0,1,1]"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(xv), "s"(STEM_PAIR(w, 0)), "s"(STEM_PAIR(w, 1)), "s"(STEM_PAIR(w, 2)), "s"(STEM_PAIR(w, 3)), "s"(STEM_PAIR(w, 4)),
          "s"(STEM_PAIR(w, 5)), "s"(STEM_PAIR(w, 6)), "s"(STEM_PAIR(w, 7)));
  else
    asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %8, %10, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %8, %11, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %3, %8, %12, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %4, %8, %13, %4 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %5, %8, %14, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %6, %8, %15, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %7, %8, %16, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(xv), "s"(STEM_PAIR(w, 0)), "s"(STEM_PAIR(w, 1)), "s"(STEM_PAIR(w, 2)), "s"(STEM_PAIR(w, 3)), "s"(STEM_PAIR(w, 4)),
          "s"(STEM_PAIR(w, 5)), "s"(STEM_PAIR(w, 6)), "s"(STEM_PAIR(w, 7)));
}

// ... the small-batch form: a tap's 16 weights as four uniform ds_read_b128 (every lane reads the same address: a broadcast) into
// VGPR pairs.  LDS reads retire in order and a read costs ~100 clk, so one tap ahead is enough; the scalar loads above are a cache
// miss per tap on a cold CU (~0.25 us each, 98 in a row: the tracker's B = 1 call was 25 us of exactly that).
struct W16v {
  f4e q[4];
};
template <int OFF>
__device__ __forceinline__ void lds_w16(W16v& w, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
               : "=&v"(w.q[0]), "=&v"(w.q[1]), "=&v"(w.q[2]), "=&v"(w.q[3])
               : "v"(addr), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48));
}
__device__ __forceinline__ void lds_wait(W16v& w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w.q[0]), "+v"(w.q[1]), "+v"(w.q[2]), "+v"(w.q[3])); }
template <int HALF>
__device__ __forceinline__ void fma8v(float2v (&acc)[NPAIR], float2v xv, const W16v& w) {
#define STEM_VP(k) __builtin_shufflevector(w.q[(k) >> 1], w.q[(k) >> 1], 2 * ((k) & 1), 2 * ((k) & 1) + 1)
  if constexpr (HALF == 0)
    asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %8, %10, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %2, %8, %11, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %8, %12, %3 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %4, %8, %13, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %5, %8, %14, %5 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %6, %8, %15, %6 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %7, %8, %16, %7 op_sel_hi:[0,1,1]"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(xv), "v"(STEM_VP(0)), "v"(STEM_VP(1)), "v"(STEM_VP(2)), "v"(STEM_VP(3)), "v"(STEM_VP(4)), "v"(STEM_VP(5)), "v"(STEM_VP(6)), "v"(STEM_VP(7)));
  else
    asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %8, %10, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %8, %11, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %3, %8, %12, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %4, %8, %13, %4 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %5, %8, %14, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %6, %8, %15, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %7, %8, %16, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(xv), "v"(STEM_VP(0)), "v"(STEM_VP(1)), "v"(STEM_VP(2)), "v"(STEM_VP(3)), "v"(STEM_VP(4)), "v"(STEM_VP(5)), "v"(STEM_VP(6)), "v"(STEM_VP(7)));
#undef STEM_VP
}

template <int I = 0, class F>
__device__ __forceinline__ void sfor7(F&& f) {
  if constexpr (I < KS) {
    f(std::integral_constant<int, I>{});
    sfor7<I + 1>(static_cast<F&&>(f));
  }
}
// PR = pooled rows per strip (template parameter of the kernel: 4, or 1 for small batches where 4-row strips leave the chip empty);
// NR = 2 PR + 1 conv rows a strip computes: 2 pa - 1 .. 2 pa + 2 PR - 1
}  // namespace stem

// The loop nest is (input channel, ky) [runtime, 14 trips] x kx [7] x conv row [9]: a tap's 16 weights are loaded ONCE
// (s_load_dwordx16, the next tap's in flight meanwhile) and feed 9 x 8 packed FMAs; the 9 input rows a (ci, ky) touches are
// loaded as one 8-byte pair per lane each and expanded to the 7 horizontal taps by DPP.
template <bool NHWC, int PR>
__global__ __launch_bounds__(HDN_BLOCK, 2) void trunk_stem_kernel(const float* __restrict__ x, const float* __restrict__ wT,
                                                                  const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                                  int Hc, int Wc, int Hp, int Wp, int strips_per_img, int total_waves) {
  using namespace stem;
  constexpr int NR = 2 * PR + 1;
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane(blockIdx.x * (HDN_BLOCK / 64) + (threadIdx.x >> 6));
  if (g >= total_waves) return;
  const int cb = g & 3, s = (g >> 2) % strips_per_img, b = (g >> 2) / strips_per_img;
  const int pa = s * PR;
  const int r0 = 2 * pa - 1;  // first conv row of the strip (row -1 does not exist: it stays 0)
  const float* __restrict__ xb = x + size_t(b) * 2 * H * W;
  const int HW = H * W;
  // own pair = columns 2c, 2c+1; a pair that would start at the row's last float is read one float earlier (nothing beyond the
  // tensor is touched) and shifted
  const bool ok0 = 2 * lane < W, ok1 = 2 * lane + 1 < W;
  const uint32_t ix0 = 4u * (uint32_t)min(2 * lane, max(W - 2, 0));
  const bool sh = W >= 2 && 2 * lane == W - 1;
  typedef float2v f2u __attribute__((aligned(4)));

  float2v acc[NR][NPAIR];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) acc[i][k] = float2v{0.f, 0.f};

  // the NR input rows of step t = (input channel, ky): iy = 2 (r0 + i) - 3 + ky, each as the four pairs its 7 taps come from:
  //   kx: 0 -> (c-2).y   1, 2 -> (c-1).x, .y   3, 4 -> own .x, .y   5, 6 -> (c+1).x, .y
  // All NR loads in flight at once, branch-free: rows beyond the image re-read row 0 and are dropped.  With few waves per SIMD (the
  // tracker's B = 1 call) a step would begin with a full HBM / L2 round trip, fourteen times over: the small-batch form below asks for
  // every step's rows before the first FMA.
  auto load_rows = [&](int t, float2v (&raw)[NR], bool (&rk)[NR]) {
    const int ci = t >= KS ? 1 : 0, ky = t - KS * ci;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int iy = 2 * (r0 + i) - 3 + ky;          // wave-uniform
      const bool rok = iy >= 0 && iy < H;
      rk[i] = rok;
      raw[i] = *reinterpret_cast<const f2u*>(reinterpret_cast<const char*>(xb + ci * HW + (rok ? iy : 0) * W) + ix0);
    }
  };
  // one step: the 7 taps of (input channel, ky) on the NR rows
  auto do_step = [&](int t, const float2v (&raw)[NR], const bool (&rk)[NR]) {
    const int ci = t >= KS ? 1 : 0, ky = t - KS * ci;
    const cfloat* wr = opaque(wT + cb * CB + (ci * KS + ky) * KS * CO);
    f16v w = sload16<0>(wr);
    float2v xv[NR][4];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const float2v v = raw[i];
      const float2v own = float2v{ok0 && rk[i] ? (sh ? v.y : v.x) : 0.f, ok1 && rk[i] ? v.y : 0.f};  // zero padding: selected, not multiplied
      xv[i][1] = float2v{from_prev_lane(own.x), from_prev_lane(own.y)};
      xv[i][0] = float2v{0.f, from_prev_lane(xv[i][1].y)};
      xv[i][2] = own;
      xv[i][3] = float2v{from_next_lane(own.x), from_next_lane(own.y)};
    }
    swait(w);
    sfor7([&](auto KX) {
      constexpr int kx = decltype(KX)::value;
      f16v wn;
      if constexpr (kx + 1 < KS) wn = sload16<(kx + 1) * CO * 4>(wr);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        if constexpr ((kx + 1) & 1) fma8<1>(acc[i], xv[i][(kx + 1) >> 1], w);
        else fma8<0>(acc[i], xv[i][(kx + 1) >> 1], w);
      }
      if constexpr (kx + 1 < KS) {
        swait(wn);
        w = wn;
      }
    });
  };
  if constexpr (PR == 1) {
    // small batches (few waves per SIMD, nothing to hide a round trip behind): ALL 14 x 3 input rows are asked for up front
    // (84 registers), the loop is unrolled
    float2v raw[2 * KS][NR];
    bool rk[2 * KS][NR];
#pragma unroll
    for (int t = 0; t < 2 * KS; ++t) load_rows(t, raw[t], rk[t]);
    // this wave's 98 x 16 weights -> its LDS table [tap][16] (one round trip, beside the rows')
    constexpr int WTAB = 2 * KS * KS * CB;                       // floats
    __shared__ __attribute__((aligned(16))) float sw[HDN_BLOCK / 64][WTAB];
    float* const tab = sw[threadIdx.x >> 6];
    {
      constexpr int N4 = WTAB / 4, WIT = cdiv(N4, 64);
      f4e wv[WIT];
#pragma unroll
      for (int q = 0; q < WIT; ++q) {
        const int i4 = min(lane + q * 64, N4 - 1), tap = i4 >> 2, k4 = i4 & 3;
        wv[q] = *reinterpret_cast<const f4e*>(wT + size_t(tap) * CO + cb * CB + k4 * 4);
      }
#pragma unroll
      for (int q = 0; q < WIT; ++q)
        if (lane + q * 64 < N4) reinterpret_cast<f4e*>(tab)[lane + q * 64] = wv[q];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table is written (a wave reads only its own table: no barrier)
    const uint32_t tab_addr = lds_addr(tab);
    auto do_step_lds = [&](int t, const float2v (&rw)[NR], const bool (&rkk)[NR]) {
      const uint32_t wa = tab_addr + t * (KS * CB * 4);
      W16v w;
      lds_w16<0>(w, wa);
      float2v xv[NR][4];
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const float2v v = rw[i];
        const float2v own = float2v{ok0 && rkk[i] ? (sh ? v.y : v.x) : 0.f, ok1 && rkk[i] ? v.y : 0.f};
        xv[i][1] = float2v{from_prev_lane(own.x), from_prev_lane(own.y)};
        xv[i][0] = float2v{0.f, from_prev_lane(xv[i][1].y)};
        xv[i][2] = own;
        xv[i][3] = float2v{from_next_lane(own.x), from_next_lane(own.y)};
      }
      lds_wait(w);
      sfor7([&](auto KX) {
        constexpr int kx = decltype(KX)::value;
        W16v wn;
        if constexpr (kx + 1 < KS) lds_w16<(kx + 1) * CB * 4>(wn, wa);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          if constexpr ((kx + 1) & 1) fma8v<1>(acc[i], xv[i][(kx + 1) >> 1], w);
          else fma8v<0>(acc[i], xv[i][(kx + 1) >> 1], w);
        }
        if constexpr (kx + 1 < KS) {
          lds_wait(wn);
          w = wn;
        }
      });
    };
#pragma unroll
    for (int t = 0; t < 2 * KS; ++t) do_step_lds(t, raw[t], rk[t]);
  } else {
    // large batches: enough waves per SIMD to hide the rows' round trip; asking for step t + 1's rows a step ahead costs registers
    // and measured 48 -> 53 us at B = 64
#pragma unroll 1
    for (int t = 0; t < 2 * KS; ++t) {
      float2v raw[NR];
      bool rk[NR];
      load_rows(t, raw, rk);
      do_step(t, raw, rk);
    }
  }

  // + bias, ReLU; conv rows / columns that do not exist are 0 (they never win a maximum: every window has a real element >= 0)
  const cfloat* bq = opaque(bias + cb * CB);
  const bool lane_ok = lane < Wc;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const bool keep = lane_ok && r0 + i >= 0 && r0 + i < Hc;
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const float t0 = acc[i][k].x + bq[2 * k], t1 = acc[i][k].y + bq[2 * k + 1];
      acc[i][k] = float2v{keep ? fmaxf(t0, 0.f) : 0.f, keep ? fmaxf(t1, 0.f) : 0.f};
    }
  }
  // 3 x 3 / stride 2 maxima: vertical over conv rows 2p-1, 2p, 2p+1 = strip rows 2j, 2j+1, 2j+2; horizontal over lanes c-1, c, c+1
#pragma unroll
  for (int jp = 0; jp < PR; ++jp) {
    const int p = pa + jp;
    float res[CB];
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[2 * jp][k].x, acc[2 * jp + 1][k].x), acc[2 * jp + 2][k].x);
      const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[2 * jp][k].y, acc[2 * jp + 1][k].y), acc[2 * jp + 2][k].y);
      res[2 * k] = __builtin_fmaxf(__builtin_fmaxf(from_prev_lane(m0), m0), from_next_lane(m0));
      res[2 * k + 1] = __builtin_fmaxf(__builtin_fmaxf(from_prev_lane(m1), m1), from_next_lane(m1));
    }
    const int q = lane >> 1;
    if ((lane & 1) == 0 && q < Wp && p < Hp) {
      if constexpr (NHWC) {
        f4e* o = reinterpret_cast<f4e*>(out + ((size_t(b) * Hp + p) * Wp + q) * CO + cb * CB);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = f4e{res[4 * k], res[4 * k + 1], res[4 * k + 2], res[4 * k + 3]};
      } else {
#pragma unroll
        for (int k = 0; k < CB; ++k) out[((size_t(b) * CO + cb * CB + k) * Hp + p) * Wp + q] = res[k];
      }
    }
  }
}

}  // namespace hdn

extern "C" int hdn_trunk_stem_f32(const float* x, const float* wT, const float* bias, float* out, int B, int H, int W, int nhwc,
                                  void* stream) {
  if (!x || !wT || !bias || !out) return HDN_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0) return HDN_E_SHAPE;
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
  if (W < 2 || Wc > 64 || B > 65535 || (long long)H * W > (1LL << 24)) return HDN_E_LIMIT;  // a lane reads its columns as a pair; one lane per conv column
  if (out == x) return HDN_E_ALIAS;
  // a wave: PR pooled rows x 16 channels x the full width.  4-row strips share most of their input rows (9 conv rows for 4 pooled
  // ones); below ~1,000 waves (B < 32 at 127 px) the chip is better filled by 1-row strips (3 conv rows each: 1.3 x the arithmetic,
  // 4 x the waves, a quarter of the chain per wave) - the tracker's B = 1 call drops from 35 to 27 us, and to 12 us with all of a wave's 42 input rows asked for up front (see the kernel)
  const bool small = (long long)B * 4 * hdn::cdiv(Hp, 4) < 1024;
  const int pr = small ? 1 : 4;
  const int strips = hdn::cdiv(Hp, pr);
  const long long total = (long long)B * 4 * strips;
  if (total > 0x7fffffffLL) return HDN_E_LIMIT;
  const dim3 grid((unsigned)((total + 3) / 4)), block(HDN_BLOCK);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define HDN_STEM_LAUNCH(NHWC_, PR_) \
  hipLaunchKernelGGL((hdn::trunk_stem_kernel<NHWC_, PR_>), grid, block, 0, st, x, wT, bias, out, H, W, Hc, Wc, Hp, Wp, strips, (int)total)
  if (nhwc) { if (small) HDN_STEM_LAUNCH(true, 1); else HDN_STEM_LAUNCH(true, 4); }
  else { if (small) HDN_STEM_LAUNCH(false, 1); else HDN_STEM_LAUNCH(false, 4); }
#undef HDN_STEM_LAUNCH
  return hdn::launch_status();
}
