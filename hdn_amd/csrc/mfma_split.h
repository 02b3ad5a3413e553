// What every two-fp16-piece matrix-core kernel of this library shares (conv3x3.hip, conv3x3s2.hip, trunk_stem_mfma.hip, head_conv.hip, head_tail.hip):
// the vector types of v_mfma_f32_32x32x16_f16's operands, the instruction itself, the fp32 -> two-piece split and a compile-time loop.
//
// fp32 value x is carried as x = p0 + 2^-11 p1 with p0 = fp16(x) and p1 = fp16((x - p0) * 2^11) (round-to-nearest-even by v_cvt_pk_f16_f32 each; the
// residual x - p0 is exact in fp32 and so is its product with 2^11): |x - (p0 + 2^-11 p1)| <= 2^-23 |x|.  A product of two such values is accumulated
// as p0 q0 into a "hi" fp32 accumulator and p0 q1 + p1 q0 into a "lo" one (scaled by 2^11; the p1 q1 term is below fp32's resolution): three MFMAs
// per fp32 multiply-accumulate, the error of an fp32 convolution (analysis and measurements: conv3x3.hip).  Needs |x| < 65,504
// (hdn_common.h: check_fp16_range).
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

// two fp32 values -> their two pieces, packed (low half = first value): 6 VALU operations per pair
__device__ __forceinline__ void split2(f2 v, unsigned& p0, unsigned& p1) {
  const f16x2 h = __builtin_convertvector(v, f16x2);
  p0 = __builtin_bit_cast(unsigned, h);
  const f2 r = (v - __builtin_convertvector(h, f2)) * LO_SCALE;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ void split2x2(float x, float y, unsigned& p0, unsigned& p1) { split2(f2{x, y}, p0, p1); }

}  // namespace mc
}  // namespace hdn
