// What every two-fp16-piece matrix-core kernel of this library shares (conv3x3.hip, conv3x3s2.hip, trunk_stem_mfma.hip, head_conv.hip, head_tail.hip):
// the vector types of v_mfma_f32_32x32x16_f16's operands, the instruction itself, the fp32 -> two-piece split and a compile-time loop.
//
// fp32 value x is carried as x = p0 + 2^-11 p1 with p0 = fp16(x) and p1 = fp16((x - p0) * 2^11) (round-to-nearest-even by v_cvt_pk_f16_f32 each; the
// residual x - p0 is exact in fp32 and so is its product with 2^11): |x - (p0 + 2^-11 p1)| <= 2^-23 |x|.  A product of two such values is accumulated
// as p0 q0 into a "hi" fp32 accumulator and p0 q1 + p1 q0 into a "lo" one (scaled by 2^11; the p1 q1 term is below fp32's resolution): three MFMAs
// per fp32 multiply-accumulate, the error of an fp32 convolution (analysis and measurements: conv3x3.hip).
//
// Range.  fp16 ends at 65,504; the reference's fp32 convolutions do not.  ACTIVATIONS (everything split in a kernel) are therefore split as
// x 2^-8 and the accumulators joined as (hi + 2^-11 lo) 2^8 (join() below; powers of two: exact), which moves the limit to |x| < 65,520 x 256 =
// 1.67e7 for one more packed multiply per staged pair and one more multiply per output.  What it costs in precision: the second piece of an
// activation below 2^-14 x 2^8 = 0.0156 becomes an fp16 subnormal, an ABSOLUTE error of at most 2^-36 x 2^8 = 3.7e-9 per activation instead of a
// relative 2^-23 — below the rounding of the fp32 sums these values enter.  WEIGHTS are split by the packers on the host, unscaled, and checked
// there (|w| < 65,504 or the packer raises).  Beyond 1.67e7 the result is inf / NaN as before; HDN_CHECK_RANGE=1 (range_check.hip) reports it.
// The multiply is paid once per TRUNK, not once per layer: the entry points take `act_domain` (0: activations in real units in and out; 1: in and out are
// x 2^-8), hdn_amd.trunk enters the scaled domain at the first stage's output and leaves it at the pooled regressor (hdn_avgpool_fc_f32 multiplies by 2^8),
// biases are handed over pre-scaled (exact) — inside, the kernels are the 6-operation split of ABI <= 8 with the range of ABI 9.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace hdn {
namespace mc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr float LO_SCALE = 2048.f, LO_UNSCALE = 1.f / 2048.f;
#ifndef HDN_ACT_SCALE_LOG2
#define HDN_ACT_SCALE_LOG2 8                                       // (0: the pre-ABI-9 split, for the A/B build of tools/experiments/ab_act_scale.sh)
#endif
constexpr float ACT_UNSCALE = float(1 << HDN_ACT_SCALE_LOG2), ACT_SCALE = 1.f / ACT_UNSCALE;   // activations are split as x * ACT_SCALE (see "Range" above)

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N - 1
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// D (32 x 32 fp32, 16 registers: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) += A (lane = (row, k half), 8 fp16) x B (lane = (k half, column), 8 fp16)
__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two fp32 values -> their two pieces, packed (low half = first value): 7 VALU operations per pair, 6 in the scaled domain.
// SD ("scaled domain", the fused trunk's interior): the value in memory is already x * ACT_SCALE — no multiply here, and join<true> leaves the sum scaled.
template <bool SD = false>
__device__ __forceinline__ void split2(f2 v, unsigned& p0, unsigned& p1) {
  if constexpr (!SD) v = v * ACT_SCALE;
  const f16x2 h = __builtin_convertvector(v, f16x2);
  p0 = __builtin_bit_cast(unsigned, h);
  const f2 r = (v - __builtin_convertvector(h, f2)) * LO_SCALE;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
template <bool SD = false>
__device__ __forceinline__ void split2x2(float x, float y, unsigned& p0, unsigned& p1) { split2<SD>(f2{x, y}, p0, p1); }

// the fp32 sum of a "hi" and a "lo" accumulator of split activations: (hi + 2^-11 lo) 2^8, one rounding (the two scalings are exact)
// (SD: the result stays in the scaled domain, hi + 2^-11 lo)
template <bool SD = false>
__device__ __forceinline__ float join(float hi, float lo) {
  if constexpr (SD) return hi + lo * LO_UNSCALE;
  else return hi * ACT_UNSCALE + lo * (LO_UNSCALE * ACT_UNSCALE);
}

}  // namespace mc
}  // namespace hdn
