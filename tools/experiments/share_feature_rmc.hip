// EXPERIMENT (round 5, not built into the library; measured and dropped: profiles/round5_conv3x3.txt section 6).  To try it again: copy to
// hdn_amd/csrc/, add it to HIP_SOURCES in __graft_entry__.py and call launch_sf_rmc from hdn_share_feature_f32 for W <= 128.
// PreShareFeature (eval mode), rows in registers, with its middle layer on the matrix cores (many images of width <= 128):
//   3 x (conv3x3 pad 1, no bias -> BatchNorm(running stats) -> ReLU), channels 1 -> 4 -> 8 -> 1.
// Reference: homo_estimator/Deep_homography/Oneline_DLTv1/preprocess/input_feature_extractor.py:3-29.
// Same data flow as share_feature_rows_kernel (share_feature.hip): one WAVE owns a strip of output rows over the full width, lane l holds the pixel
// pair (2l, 2l + 1), nothing goes through the LDS.  The 4 -> 8 layer (72 % of the multiply-adds: 288 packed FMAs per row and wave there) is a GEMM on
// v_mfma_f32_32x32x16_f16 here, fp32 as two fp16 pieces (mfma_split.h), with the WEIGHTS as the A operand and the pixels as B, so that the result
// comes back pixel-major, i.e. in the layout the lanes already have:
//   row m    = (output channel co, pixel of the pair dx, row of a row PAIR dy) = 8 x 2 x 2 = 32,
//   column n = a pixel pair (32 per MFMA tile: two tiles per row),
//   k        = (ky4, input channel, kx4): the 4 x 4 window of layer-1 pixels under the 2 x 2 outputs, 4 x 4 x 4 = 64 = four k steps (one per window row);
//              A[m][k] = w2[co][ci][ky4 - dy][kx4 - dx], zero where that is not a tap.
// B fragment of a lane = (its pair, two of the four channels) x the 4 window columns 2l - 1 .. 2l + 2 of ONE layer-1 row: values the lane computed itself
// plus one from each neighbour (two DPP shifts per channel), split in registers.  The k halves of an MFMA tile want the other two channels from the lane
// 32 further on, and the accumulator tile comes back with half of the 32 rows in that lane again: one v_permlane32_swap per register each way.
#include <cstdlib>

#include "hdn_common.h"
#include "mfma_split.h"

namespace hdn {
namespace sfr {
using namespace hdn::mc;

constexpr int SF_W1 = 0, SF_W2 = 36, SF_W3 = 324, SF_ALPHA = 396, SF_BETA = 409;   // the parameter block (include/hdn_hip.h)

__device__ __forceinline__ float from_prev_lane(float v) {  // lane l <- lane l-1, lane 0 <- 0   (wave_shr:1, bound_ctrl)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_next_lane(float v) {  // lane l <- lane l+1, lane 63 <- 0  (wave_shl:1, bound_ctrl)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// A wave-uniform pointer into the parameter block in the constant address space (scalar loads), opaque to the optimiser: a layer's weights are loaded
// where they are used instead of all 396 being hoisted out of the row loop and spilled (the first build moved 372 values per iteration through
// v_readlane / v_writelane).
typedef const float __attribute__((address_space(4))) cfloat;
__device__ __forceinline__ const cfloat* opaque(const float* p) {
  uint64_t a = reinterpret_cast<uint64_t>(p);
  asm volatile("" : "+s"(a));
  return (const cfloat*)a;
}
typedef const f2 __attribute__((address_space(4), aligned(8))) cf2;
// acc += v * w.x / acc += v * w.y for both pixels of v: one half of an SGPR PAIR broadcast by the operand selects (the compiler's own form keeps a
// (w, w) pair per weight: twice the scalar registers, spilled through v_writelane / v_readlane)
__device__ __forceinline__ void fma_lo(f2& acc, f2 v, f2 w) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(v), "s"(w)); }
__device__ __forceinline__ void fma_hi(f2& acc, f2 v, f2 w) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(v), "s"(w)); }
// lanes 32..63 of a <-> lanes 0..31 of b
// (the instruction needs 2 wait states behind a VALU write of an operand; the compiler does not look into asm statements)
__device__ __forceinline__ void half_swap(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void half_swap(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

struct Frag {            // a layer-1 row as the B operands of the two MFMA tiles of a row: [tile][piece]
  u32x4 t[2][2];
};

__global__ __launch_bounds__(HDN_BLOCK) void share_feature_rmc_kernel(const float* __restrict__ img, const float* __restrict__ prm, float* __restrict__ out,
                                                                      int H, int W, int rows_per_strip, int strips_per_img, int total_strips) {
  const int lane = threadIdx.x & 63;
  const int strip = __builtin_amdgcn_readfirstlane(blockIdx.x * (HDN_BLOCK / 64) + (threadIdx.x >> 6));
  if (strip >= total_strips) return;
  const int bimg = strip / strips_per_img, ks = strip - bimg * strips_per_img;
  const float* __restrict__ src = img + size_t(bimg) * H * W;
  float* __restrict__ dst = out + size_t(bimg) * H * W;
  const int ra = ks * rows_per_strip, rb = min(ra + rows_per_strip, H);
  const bool m0 = 2 * lane < W, m1 = 2 * lane + 1 < W;
  const f2 cm = {m0 ? 1.f : 0.f, m1 ? 1.f : 0.f};
  const int col0 = m0 ? 2 * lane : 0, col1 = m1 ? 2 * lane + 1 : 0;

  // ---- layer 2's weights as A fragments: lane = (m = lane & 31 = (co, dy, dx), k half g = channels 2g, 2g + 1); k step s = window row ky4
  u32x4 wa[4][2];
  {
    const int m = lane & 31, g = lane >> 5, dx = m & 1, dy = (m >> 1) & 1, co = m >> 2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ky = s - dy;
      unsigned h[4], l[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = 2 * jj + e, ci = 2 * g + (j >> 2), kx = (j & 3) - dx;
          const bool tap = ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2;
          v[e] = tap ? prm[SF_W2 + (((ci >> 1) * 9 + (tap ? ky * 3 + kx : 0)) * 8 + co) * 2 + (ci & 1)] : 0.f;
        }
        split2x2(v[0], v[1], h[jj], l[jj]);
      }
      wa[s][0] = u32x4{h[0], h[1], h[2], h[3]};
      wa[s][1] = u32x4{l[0], l[1], l[2], l[3]};
    }
  }

  auto load_row = [&](int i) -> f2 {                           // input row i at this lane's two columns (zero outside the image)
    const bool ok = i >= 0 && i < H;
    const float* p = src + (ok ? i : 0) * W;
    const float x0 = p[col0], x1 = p[col1];
    return f2{ok && m0 ? x0 : 0.f, ok && m1 ? x1 : 0.f};
  };

  // layer-1 row k from input rows k - 1, k, k + 1 -> its B fragments.  in[r] = the three input rows.
  auto layer1 = [&](int k, const f2 (&in)[3], Frag& fr) {
    const bool row_ok = k >= 0 && k < H;                       // (outside the image: the next convolution's zero padding)
    const cfloat* const wp = opaque(prm);
    f2 win[3][3];                                              // [row][kx]: (value at col0 + kx - 1, value at col1 + kx - 1)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      win[r][1] = in[r];
      win[r][0] = f2{from_prev_lane(in[r].y), in[r].x};
      win[r][2] = f2{in[r].y, from_next_lane(in[r].x)};
    }
    unsigned xh[2][4], xl[2][4];                               // [channel pair g][dword]: channel 2g: dwords 0, 1; channel 2g + 1: dwords 2, 3
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const cf2* const w1 = reinterpret_cast<const cf2*>(opaque(prm) + SF_W1);     // w1t[tap][channel]: (channel 2g, 2g + 1) = pair 2 tap + g
      f2 acc[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f2 w = w1[(ky * 3 + kx) * 2 + g];
          fma_lo(acc[0], win[ky][kx], w);
          fma_hi(acc[1], win[ky][kx], w);
        }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ch = 2 * g + c;
        const float al = wp[SF_ALPHA + ch], be = wp[SF_BETA + ch];
        f2 y = __builtin_elementwise_fma(acc[c], f2{al, al}, f2{be, be});
        y.x = row_ok && m0 ? fmaxf(y.x, 0.f) : 0.f;
        y.y = row_ok && m1 ? fmaxf(y.y, 0.f) : 0.f;
        const float left = from_prev_lane(y.y), right = from_next_lane(y.x);     // columns 2l - 1, 2l + 2
        split2x2(left, y.x, xh[g][2 * c], xl[g][2 * c]);
        split2x2(y.y, right, xh[g][2 * c + 1], xl[g][2 * c + 1]);
      }
    }
    // tile 0 (pairs 0..31): lanes < 32 give their channels 0, 1, lanes >= 32 the channels 2, 3 of the lane 32 back; tile 1 the other way round
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      half_swap(xh[0][d], xh[1][d]);
      half_swap(xl[0][d], xl[1][d]);
    }
    fr.t[0][0] = u32x4{xh[0][0], xh[0][1], xh[0][2], xh[0][3]};
    fr.t[0][1] = u32x4{xl[0][0], xl[0][1], xl[0][2], xl[0][3]};
    fr.t[1][0] = u32x4{xh[1][0], xh[1][1], xh[1][2], xh[1][3]};
    fr.t[1][1] = u32x4{xl[1][0], xl[1][1], xl[1][2], xl[1][3]};
  };

  Frag fr[4];                                                  // layer-1 rows j - 1 .. j + 2 of the current row pair (j, j + 1)
  f2 inr[3];                                                   // input rows around the next layer-1 row
  f2 P[4][3];                                                  // layer-3 partial sums: [output row j - 1 .. j + 2][kx], at the column the layer-2 value sits at
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) P[o][kx] = f2{0.f, 0.f};

  int nk = ra - 2;                                             // next layer-1 row to make
  inr[0] = load_row(nk - 1);
  inr[1] = load_row(nk);
  inr[2] = load_row(nk + 1);
  auto next_l1 = [&](Frag& f) {
    layer1(nk, inr, f);
    ++nk;
    inr[0] = inr[1];
    inr[1] = inr[2];
    inr[2] = load_row(nk + 1);
  };
  next_l1(fr[0]);
  next_l1(fr[1]);

#pragma unroll 1
  for (int j = ra - 1; j < rb + 1; j += 2) {                   // layer-2 rows j, j + 1 from layer-1 rows j - 1 .. j + 2
    next_l1(fr[2]);
    next_l1(fr[3]);
    f32x16 dh[2], dl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[t][r] = dl[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        dl[t] = mfma(wa[s][1], fr[s].t[t][0], dl[t]);
        dh[t] = mfma(wa[s][0], fr[s].t[t][0], dh[t]);
        dl[t] = mfma(wa[s][0], fr[s].t[t][1], dl[t]);
      }
    }
    // own pair's 32 values: after the swap register r of v[0] is row m = (r & 3) + 8 (r >> 2), of v[1] row m + 4, m = 4 co + 2 dy + dx
    float v[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[0][r] = dh[0][r] + dl[0][r] * LO_UNSCALE;
      v[1][r] = dh[1][r] + dl[1][r] * LO_UNSCALE;
      half_swap(v[0][r], v[1][r]);
    }
    // BN + ReLU, zero outside the image, and the scatter into layer 3's partial sums: layer-2 row j + dy feeds output rows j + dy + 1 - ky
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      if (j + dy < 0 || j + dy >= H) continue;                 // (a layer-2 row outside the image is layer 3's zero padding: nothing to add)
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
        const cfloat* const wp = opaque(prm);                  // (per channel pair: 9 weight pairs + 4 BN values live at a time)
        const cf2* const w3 = reinterpret_cast<const cf2*>(wp + SF_W3) + cp * 9;      // w3p[channel pair][tap][channel of the pair]
        f2 y[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int co = 2 * cp + c, m = 4 * co + 2 * dy, hsel = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3);       // m = (r & 3) + 8 (r >> 2) + 4 h
          const float al = wp[SF_ALPHA + 4 + co], be = wp[SF_BETA + 4 + co];
          y[c] = __builtin_elementwise_fma(f2{v[hsel][r], v[hsel][r + 1]}, f2{al, al}, f2{be, be});
          y[c].x = fmaxf(y[c].x, 0.f) * cm.x;                  // (columns >= W: zero; cm = 1 / 0 per pixel of the pair)
          y[c].y = fmaxf(y[c].y, 0.f) * cm.y;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const f2 w = w3[ky * 3 + kx];
            fma_lo(P[dy + 2 - ky][kx], y[0], w);               // output row (j + dy) + 1 - ky = (j - 1) + (dy + 2 - ky)
            fma_hi(P[dy + 2 - ky][kx], y[1], w);
          }
      }
    }
    // output rows j - 1 and j are complete
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int orow = j - 1 + o;
      if (orow >= ra && orow < rb) {
        asm volatile("s_nop 1");               // the partial sums come straight out of asm FMAs the compiler's DPP hazard check cannot see
        const f2 A = P[o][0], Bm = P[o][1], C = P[o][2];
        const float al = prm[SF_ALPHA + 12], be = prm[SF_BETA + 12];
        const f2 s = {from_prev_lane(A.y) + Bm.x + C.y, A.x + Bm.y + from_next_lane(C.x)};
        const f2 y = __builtin_elementwise_fma(s, f2{al, al}, f2{be, be});
        float* q = dst + size_t(orow) * W;
        if (m0) q[col0] = fmaxf(y.x, 0.f);
        if (m1) q[col1] = fmaxf(y.y, 0.f);
      }
    }
    // the ring moves on by two rows
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      P[0][kx] = P[2][kx];
      P[1][kx] = P[3][kx];
      P[2][kx] = f2{0.f, 0.f};
      P[3][kx] = f2{0.f, 0.f};
    }
    fr[0] = fr[2];
    fr[1] = fr[3];
  }
}

}  // namespace sfr

// called by hdn_share_feature_f32 (share_feature.hip) for W <= 128 and enough images to fill the chip
int launch_sf_rmc(const float* img, const float* folded, float* out, int B, int H, int W, hipStream_t stream) {
  static const int forced = [] { const char* e = getenv("HDN_SF_RMC_STRIP"); return e ? atoi(e) : 0; }();  // A/B switch: rows per wave (even)
  int n = 8;
  if (forced > 0) n = forced + (forced & 1);
  const int strips = cdiv(H, n);
  const long long total = (long long)strips * B;
  if (total > 0x7fffffffLL) return HDN_E_LIMIT;
  const int wpb = HDN_BLOCK / 64;
  hipLaunchKernelGGL(sfr::share_feature_rmc_kernel, dim3((unsigned)((total + wpb - 1) / wpb)), dim3(HDN_BLOCK), 0, stream, img, folded, out, H, W, n, strips,
                     (int)total);
  return launch_status();
}

}  // namespace hdn
