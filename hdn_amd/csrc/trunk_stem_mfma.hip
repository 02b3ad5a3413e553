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
// Workgroup = 4 waves = (image, block of 16 conv rows = 8 pooled rows).  Per iteration every wave computes ONE conv row (64 pixels x 64 channels:
// 2 x 2 MFMA tiles, 7 k steps x 12 MFMAs):
//   * the input rows of the iteration (13, of which 8 are new) sit in an LDS ring as fp32, zero-padded (3 columns left, the out-of-range rows as zeros):
//     no masking in the loop; a lane reads its 8 floats as four 8-byte ds_reads (column 2 ox: 8-byte aligned) and splits them in registers;
//   * the weights are host-packed in fragment order ([k step][n tile][piece][lane] x 16 B, 28 KB) and copied to LDS once per workgroup;
//   * the conv row (+ bias, ReLU) goes to an LDS ring of five rows [pixel][channel] fp32; after the barrier all 256 threads pool the two pooled rows the
//     iteration completes (3 x 3 windows as 16-byte reads over 4 channels; a window always holds a real element and everything is >= 0 behind the ReLU, so
//     the pool's -inf padding is a plain skip) and store them channels-last, 16 bytes per lane.
// Iteration 0 computes the block's halo rows (only conv row 16 q - 1 is used): 5 iterations for 16 useful conv rows.
#include <type_traits>
#include <utility>

#include "hdn_common.h"

namespace hdn {
namespace stem_mc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int H = 127, W = 127, HC = 64, WC = 64, HP = 32, WP = 32, CO = 64, NSTEP = 7;
constexpr int ROWS_PER_BLOCK = 16, NBLK = HC / ROWS_PER_BLOCK, NITER = ROWS_PER_BLOCK / 4 + 1;
constexpr int IN_PITCH = 136;                                  // floats per staged input row: 3 zeros + 127 pixels + 6 zeros
constexpr int IN_SLOTS = 16;                                   // input rows in the ring (13 are live)
constexpr int IN_BYTES = IN_SLOTS * 2 * IN_PITCH * 4;          // [slot][ci][IN_PITCH]
constexpr int W_WORDS = NSTEP * 2 * 2 * 64;                    // 16-byte words: [k step][n tile][piece][lane]
constexpr int W_BYTES = W_WORDS * 16;
constexpr int ROW_PITCH = CO + 4;                              // floats per pixel of a conv row in LDS
constexpr int ROW_BYTES = WC * ROW_PITCH * 4;
constexpr int ROW_SLOTS = 5;
constexpr int LDS_BYTES = W_BYTES + IN_BYTES + ROW_SLOTS * ROW_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
constexpr float LO_SCALE = 2048.f, LO_UNSCALE = 1.f / 2048.f;

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
__device__ __forceinline__ void split2x2(f2 v, unsigned& p0, unsigned& p1) {
  const f16x2 h = __builtin_convertvector(v, f16x2);
  p0 = __builtin_bit_cast(unsigned, h);
  const f2 r = (v - __builtin_convertvector(h, f2)) * LO_SCALE;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

__global__ __launch_bounds__(HDN_BLOCK) void trunk_stem_mfma_kernel(const float* __restrict__ x, const u32x4* __restrict__ wfrag, const float* __restrict__ bias,
                                                                    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* const sW = reinterpret_cast<u32x4*>(smem);
  float* const sIn = reinterpret_cast<float*>(smem + W_BYTES);
  float* const sRow = reinterpret_cast<float*>(smem + W_BYTES + IN_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, g = lane >> 5;
  const int b = blockIdx.x / NBLK, q = blockIdx.x % NBLK;
  const float* const xb = x + (size_t)b * 2 * H * W;

  for (int i = tid; i < W_WORDS; i += HDN_BLOCK) sW[i] = wfrag[i];
  // bias of this lane's two channels (C/D layout: column = lane & 31)
  const float bias0 = bias[li], bias1 = bias[32 + li];

  // input rows [y_lo, y_lo + nrow) (both channels) -> the ring: slot = row mod 16, zeros outside the image.  Work item = (row, channel, 4 padded columns)
  auto load_item = [&](int item, int y_lo) -> f4 {
    const int c4 = item % (IN_PITCH / 4), rc = item / (IN_PITCH / 4), ci = rc & 1, y = y_lo + (rc >> 1);
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < H) {
      const float* src = xb + ((size_t)ci * H + y) * W;
      const int x0 = c4 * 4 - 3;                                         // image column of the first of the four floats
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (x0 + j >= 0 && x0 + j < W) v[j] = src[x0 + j];
    }
    return v;
  };
  auto store_item = [&](int item, int y_lo, f4 v) {
    const int c4 = item % (IN_PITCH / 4), rc = item / (IN_PITCH / 4), ci = rc & 1, y = y_lo + (rc >> 1);
    *reinterpret_cast<f4*>(sIn + ((size_t)((y & (IN_SLOTS - 1)) * 2 + ci)) * IN_PITCH + c4 * 4) = v;
  };
  constexpr int NEW_ITEMS = 8 * 2 * (IN_PITCH / 4), NPRE = cdiv(NEW_ITEMS, HDN_BLOCK);       // the 8 new rows of an iteration
  f4 pre[NPRE];

  f32x16 acc[2][2], accl[2][2];
  int carry_slot = 0;                                                    // ring slot of conv row (first row of the iteration) - 1
  {
    const int y_lo = 2 * (q * ROWS_PER_BLOCK - 4) - 3;                   // iteration 0 (the halo rows): all 13 input rows
#if !(defined(HDN_ABLATION) && defined(STEM_EXP_NOSTAGE0))
    for (int item = tid; item < 13 * 2 * (IN_PITCH / 4); item += HDN_BLOCK) store_item(item, y_lo, load_item(item, y_lo));
#endif
  }
  __syncthreads();
  for (int it = 0; it < NITER; ++it) {
    const int c0 = q * ROWS_PER_BLOCK + (it - 1) * 4;                    // first conv row of the iteration; its input rows are 2 c0 - 3 .. 2 c0 + 9
    // the next iteration's 8 new input rows (2 c0 + 10 .. 2 c0 + 17) start their way to registers under this iteration's MFMAs
#if defined(HDN_ABLATION) && defined(STEM_EXP_NOPRE)
    for (int k = 0; k < NPRE; ++k) pre[k] = f4{0.f, 0.f, 0.f, 0.f};
#else
    if (it + 1 < NITER) {
#pragma unroll
      for (int k = 0; k < NPRE; ++k) {
        const int item = tid + k * HDN_BLOCK;
        if (item < NEW_ITEMS) pre[k] = load_item(item, 2 * c0 + 10);
      }
    }
#endif
    const int oy = c0 + wave;
#if defined(HDN_ABLATION) && defined(STEM_EXP_NOMFMA)
    const bool live = false;
#else
    const bool live = oy >= 0 && oy < HC && (it > 0 || wave == 3);       // (iteration 0: only the last halo row is ever read)
#endif
    if (live) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = accl[mt][nt][r] = 0.f;
      static_for<NSTEP>([&](auto Sc) {
        constexpr int s = decltype(Sc)::value;
        const int r = 2 * s + g, ci = r >= 7 ? 1 : 0, ky = r - 7 * ci;
        const int y = 2 * oy + ky - 3;
        const float* row = sIn + ((size_t)((y & (IN_SLOTS - 1)) * 2 + ci)) * IN_PITCH;
        u32x4 a[2][2], bf[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const f2* p = reinterpret_cast<const f2*>(row + 2 * (32 * mt + li));      // columns 2 ox .. 2 ox + 7 of the padded row = x_in 2 ox - 3 ..
          const f2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
          unsigned h[4], l[4];
          split2x2(v0, h[0], l[0]);
          split2x2(v1, h[1], l[1]);
          split2x2(v2, h[2], l[2]);
          split2x2(v3, h[3], l[3]);
          a[mt][0] = u32x4{h[0], h[1], h[2], h[3]};
          a[mt][1] = u32x4{l[0], l[1], l[2], l[3]};
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) bf[nt][pc] = sW[((s * 2 + nt) * 2 + pc) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) accl[mt][nt] = mfma(a[mt][1], bf[nt][0], accl[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma(a[mt][0], bf[nt][0], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) accl[mt][nt] = mfma(a[mt][0], bf[nt][1], accl[mt][nt]);
      });
      // ---- conv row (+ bias, ReLU) -> ring slot.  C/D layout: column (channel) = lane & 31, row (pixel) = (r & 3) + 8 (r >> 2) + 4 g
      float* const dst = sRow + (size_t)((carry_slot + 1 + wave) % ROW_SLOTS) * (WC * ROW_PITCH) + li;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          dst[px * ROW_PITCH] = fmaxf(acc[mt][0][r] + accl[mt][0][r] * LO_UNSCALE + bias0, 0.f);
          dst[px * ROW_PITCH + 32] = fmaxf(acc[mt][1][r] + accl[mt][1][r] * LO_UNSCALE + bias1, 0.f);
        }
    }
    __syncthreads();
    // ---- pooling: iteration it >= 1 completes pooled rows p = 8 q + 2 (it - 1) and p + 1 (conv rows c0 - 1 .. c0 + 3 = slots carry .. carry + 4)
#if defined(HDN_ABLATION) && defined(STEM_EXP_NOPOOL)
    if (it > 0 && x == nullptr) {
#else
    if (it > 0) {
#endif
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int item = tid + k * HDN_BLOCK;                              // (pooled row of the pair, pooled column, channel quad)
        const int c4 = item & 15, px = (item >> 4) & 31, pr = item >> 9;
        f4 m = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int crow = c0 - 1 + 2 * pr + dy;                           // conv row 2 p - 1 + dy
          if (crow < 0) continue;                                          // (the pool's padding row: every window holds a real element, all >= 0)
          const float* rowp = sRow + (size_t)((carry_slot + 2 * pr + dy) % ROW_SLOTS) * (WC * ROW_PITCH) + c4 * 4;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int cx = 2 * px - 1 + dx;
            if (cx < 0) continue;
            const f4 v = *reinterpret_cast<const f4*>(rowp + cx * ROW_PITCH);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
          }
        }
        const int p = q * (ROWS_PER_BLOCK / 2) + 2 * (it - 1) + pr;
        *reinterpret_cast<f4*>(out + (((size_t)b * HP + p) * WP + px) * CO + c4 * 4) = m;
      }
    }
    if (it + 1 < NITER) {                                                  // (every wave is past its A reads: the dead ring slots can be refilled)
#pragma unroll
      for (int k = 0; k < NPRE; ++k) {
        const int item = tid + k * HDN_BLOCK;
        if (item < NEW_ITEMS) store_item(item, 2 * c0 + 10, pre[k]);
      }
    }
    carry_slot = (carry_slot + 4) % ROW_SLOTS;                             // the iteration's last row becomes the next one's row - 1
    __syncthreads();                                                       // (the next iteration writes conv-row slots the pool just read)
  }
}

}  // namespace stem_mc
}  // namespace hdn

// wfrag: [7 k steps][2 n tiles][2 pieces][64 lanes = k half x 32 + n][8] fp16 (hdn_amd.trunk.pack_stem_mfma): element j of lane (g, n) of k step s is
// piece pc of w[co = tile * 32 + n][ci][ky][kx = j] with ci * 7 + ky = 2 s + g, and 0 for j = 7.
extern "C" int hdn_trunk_stem_mfma_f32(const float* x, const void* wfrag, const float* bias, float* out, int B, int H, int W, void* stream) {
  if (!x || !wfrag || !bias || !out) return HDN_E_NULL;
  if (B <= 0) return HDN_E_SHAPE;
  if (H != hdn::stem_mc::H || W != hdn::stem_mc::W || B > (1 << 20)) return HDN_E_LIMIT;       // 127-px crops only; hdn_trunk_stem_f32 takes the rest
  if (static_cast<const void*>(out) == static_cast<const void*>(x)) return HDN_E_ALIAS;
  if (!hdn::aligned16(wfrag) || !hdn::aligned16(out)) return HDN_E_LIMIT;
  static hdn::PerDeviceOnce attr;
  const int dev_ = hdn::PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hdn::stem_mc::trunk_stem_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       hdn::stem_mc::LDS_BYTES);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr.set(dev_);
  }
  hipLaunchKernelGGL(hdn::stem_mc::trunk_stem_mfma_kernel, dim3((unsigned)B * hdn::stem_mc::NBLK), dim3(HDN_BLOCK), hdn::stem_mc::LDS_BYTES,
                     static_cast<hipStream_t>(stream), x, static_cast<const hdn::stem_mc::u32x4*>(wfrag), bias, out);
  return hdn::launch_status();
}
