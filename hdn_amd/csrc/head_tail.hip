// The part of a MultiBAN / MultiCircBAN forward BEHIND the correlations, for the tracker's B = 1 call, as one launch:
//
//   hid[g]  = relu(W1[g] . feats[g] + b1[g])            g = (branch, level): the first 1x1 convolution + BatchNorm (folded) + ReLU of
//                                                       DepthwiseXCorr.head, hdn/models/head/ban.py:60-66, for the 2n (level, branch) pairs
//   out[br] = bf[br] + sum_l Wf[br][:, l] . hid[br, l]  the second 1x1 convolution, loc_scale and the (softmax-)weighted sum over the levels
//                                                       of MultiBAN.forward, ban.py:113-127 (linear, so folded into one weight matrix by the host)
//
// feats [2n, H, P] are the stacked correlation outputs (P = 25 x 25, 13 x 13 or 31 x 31 pixels, H = 256 hidden channels).  PyTorch runs this as two
// batched matrix products + a ReLU; hipBLASLt's choices for these shapes (M = 256, N = 169 ... 961, batch 6, fp32) take 58 us at P = 169 and
// 20 us at P = 625 for 0.13 / 0.49 GFLOP (tools/experiments/exp_head_gemm.py), more than the correlations and their convolutions' epilogues together.
//
// Here: a workgroup (4 waves) owns 32 pixels of ONE branch and walks over its n levels, so the weighted sum over the levels is a register
// accumulation in a fixed order (deterministic, no atomics).  Per level the first product runs on the matrix cores with fp32 carried as two fp16
// pieces (x = h0 + 2^-11 h1, three piece products into hi / lo accumulators: conv3x3.hip has the error analysis; the result has the error of an
// fp32 product): D[hidden channel][pixel] = W1 [H x H] . feats [H x 32].  A fragments (W1, split and laid out in fragment order by the host:
// hdn_amd.heads._pack_w1) come straight from L2, 16 bytes per lane, a whole level at a time in registers (asked for while the previous level's MFMAs
// drain); B fragments (the pixel tile of feats of ALL levels, split while it is staged: one round trip) from LDS images [level][piece][k step][k half]
// [pixel] x 16 B (one conflict-free ds_read_b128 per fragment).  A workgroup is H / 32 waves, wave w owns hidden channels [32 w, 32 w + 32).  The second
// product has only `om` <= 8 output rows: every lane multiplies the hid values it holds (16 channels of one pixel) with the level's Wf columns (LDS
// broadcast reads) in fp32 FMAs; the two half-waves and the waves are added at the end.  The launch is latency-bound (12 ... 62 workgroups): what counts
// is that a workgroup makes ONE global round trip for its activations and one per level for its weights.
#include <type_traits>

#include "hdn_common.h"
#include "mfma_split.h"

namespace hdn {
namespace ht {
using namespace hdn::mc;


constexpr int MAX_OUT = 8, TILE = 32;


// A fragments of HALF a level (HK = KSTEPS / 2 k steps x 2 pieces) for this wave's row tile, from the packed W1 [m tile][k step][piece][lane], 16 bytes
// per lane each.  The loads are volatile asm statements: program order = issue order, and they are NOT tracked by the compiler's s_waitcnt insertion
// (left to itself it sinks every load next to its MFMA and the K loop pays an L2 round trip per step: the first version of this kernel took 70 us).
// wait_half<PENDING>: the half has landed once at most PENDING younger loads are outstanding (loads return in order); the registers are operands so
// that the MFMAs reading them stay behind the wait.
template <int HK>
__device__ __forceinline__ void load_half(u32x4 (&a)[HK][2], const u32x4* wa) {
#pragma unroll
  for (int s = 0; s < HK; ++s)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[s][pc]) : "v"(wa + (s * 2 + pc) * 64));
}
template <int HK, int PENDING>
__device__ __forceinline__ void wait_half(u32x4 (&a)[HK][2]) {
  static_assert(HK == 4 || HK == 8, "one asm statement lists every register of the half");
  if constexpr (HK == 8)
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[4][0]),
                   "+v"(a[4][1]), "+v"(a[5][0]), "+v"(a[5][1]), "+v"(a[6][0]), "+v"(a[6][1]), "+v"(a[7][0]), "+v"(a[7][1])
                 : "n"(PENDING));
  else
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1])
                 : "n"(PENDING));
}

// H hidden channels, N levels (compile-time: every load of the prologue is then issued before the first wait); workgroup = H / 32 waves, one 32-row tile
// of hidden channels each
template <int H, int N>
__global__ __launch_bounds__(2 * H) void head_tail_kernel(const float* __restrict__ feats, const u32x4* __restrict__ w1p, const float* __restrict__ b1,
                                                          const float* __restrict__ wf, const float* __restrict__ bf, float* __restrict__ out, int P, int om) {
  constexpr int n = N;
  constexpr int NW = H / 32, NT = 64 * NW, KSTEPS = H / 16;
  constexpr int KH_BYTES = TILE * 16, KSTEP_BYTES = 2 * KH_BYTES, PIECE_BYTES = KSTEPS * KSTEP_BYTES, IMG_BYTES = 2 * PIECE_BYTES;   // B image of a level
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sB = smem;                                         // [level][piece][k step][k half][pixel] x 16 B
  float* const b1s = reinterpret_cast<float*>(smem + (size_t)n * IMG_BYTES);   // [level][H]
  float* const wfs = b1s + n * H;                                         // [om][level][H] (the branch's block of Wf as it lies in memory)
  float* const red = wfs + n * om * H;                                    // [wave][MAX_OUT][TILE]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, g = lane >> 5;
  const int p0 = blockIdx.x * TILE, br = blockIdx.y;

  // ---- the first level's A fragments are asked for before anything else ...
  constexpr int HK = KSTEPS / 2, HALF_LOADS = HK * 2;
  constexpr size_t LEVEL_WORDS = (size_t)NW * KSTEPS * 2 * 64, HALF_WORDS = (size_t)HK * 2 * 64;
  u32x4 a0[HK][2], a1[HK][2];                                             // k steps 0 .. HK - 1 / HK .. KSTEPS - 1 of the level in flight
  const u32x4* wa0 = w1p + (((size_t)(br * n) * NW + wave) * KSTEPS * 2) * 64 + lane;   // w1p[group][m tile][k step][piece][lane]
  load_half<HK>(a0, wa0);
  load_half<HK>(a1, wa0 + HALF_WORDS);
  // ---- ... then, all in flight together (ONE round trip): the pixel tile of feats for all levels, b1 and the branch's block of Wf
  constexpr int ITEMS = (4 * H) / NT;                                     // (pixel, 8 consecutive channels) items per thread and level
  float v[N][ITEMS][8];
#pragma unroll
  for (int l = 0; l < N; ++l)
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int item = tid + q * NT, p = item & 31, cg = item >> 5;
      const float* src = feats + (size_t)(br * n + l) * H * P + (size_t)(cg * 8) * P + min(p0 + p, P - 1);    // (clamped, not branched)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[l][q][j] = src[(size_t)j * P];
    }
  constexpr int B1_4 = N * H / 4, WF_ITERS = (MAX_OUT * N * H / 4 + NT - 1) / NT;
  const f4* b1g = reinterpret_cast<const f4*>(b1 + (size_t)br * n * H);
  const f4* wfg = reinterpret_cast<const f4*>(wf + (size_t)br * om * n * H);
  const int wf4 = om * n * H / 4;
  const f4 b1r = b1g[min(tid, B1_4 - 1)];
  f4 wfr[WF_ITERS];
#pragma unroll
  for (int q = 0; q < WF_ITERS; ++q) wfr[q] = wfg[min(tid + q * NT, wf4 - 1)];
  // split and store
#pragma unroll
  for (int l = 0; l < N; ++l)
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int item = tid + q * NT, p = item & 31, cg = item >> 5;
      const bool ok = p0 + p < P;
      unsigned q0[4], q1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split2x2(ok ? v[l][q][2 * j] : 0.f, ok ? v[l][q][2 * j + 1] : 0.f, q0[j], q1[j]);
      unsigned char* dst = sB + (size_t)l * IMG_BYTES + (cg >> 1) * KSTEP_BYTES + (cg & 1) * KH_BYTES + p * 16;
      *reinterpret_cast<u32x4*>(dst) = u32x4{q0[0], q0[1], q0[2], q0[3]};
      *reinterpret_cast<u32x4*>(dst + PIECE_BYTES) = u32x4{q1[0], q1[1], q1[2], q1[3]};
    }
  if (tid < B1_4) reinterpret_cast<f4*>(b1s)[tid] = b1r;
#pragma unroll
  for (int q = 0; q < WF_ITERS; ++q)
    if (tid + q * NT < wf4) reinterpret_cast<f4*>(wfs)[tid + q * NT] = wfr[q];
  __syncthreads();

  float part[MAX_OUT];
#pragma unroll
  for (int o = 0; o < MAX_OUT; ++o) part[o] = 0.f;
  for (int l = 0; l < n; ++l) {
    // ---- first product on the matrix cores: this wave's 32 hidden channels x the 32 pixels
    f32x16 hi, lo;
#pragma unroll
    for (int r = 0; r < 16; ++r) hi[r] = lo[r] = 0.f;
    const unsigned char* bimg = sB + (size_t)l * IMG_BYTES + g * KH_BYTES + li * 16;
    const bool more = l + 1 < n;                                          // (wave-uniform)
    const u32x4* wnext = wa0 + (size_t)(l + 1) * LEVEL_WORDS;
    // Half a level is computed while the other half and then the next level's first half travel: in flight at a wait are this half and ONE younger
    // half (an MFMA has read its operands long before a load into the same registers returns).
    wait_half<HK, HALF_LOADS>(a0);
#pragma unroll
    for (int s = 0; s < HK; ++s) {
      const u32x4 b0 = *reinterpret_cast<const u32x4*>(bimg + s * KSTEP_BYTES), b1v = *reinterpret_cast<const u32x4*>(bimg + s * KSTEP_BYTES + PIECE_BYTES);
      lo = mfma(a0[s][1], b0, lo);
      hi = mfma(a0[s][0], b0, hi);
      lo = mfma(a0[s][0], b1v, lo);
    }
    if (more) {
      load_half<HK>(a0, wnext);
      wait_half<HK, HALF_LOADS>(a1);
    } else {
      wait_half<HK, 0>(a1);
    }
#pragma unroll
    for (int s = 0; s < HK; ++s) {
      const u32x4 b0 = *reinterpret_cast<const u32x4*>(bimg + (HK + s) * KSTEP_BYTES), b1v = *reinterpret_cast<const u32x4*>(bimg + (HK + s) * KSTEP_BYTES + PIECE_BYTES);
      lo = mfma(a1[s][1], b0, lo);
      hi = mfma(a1[s][0], b0, hi);
      lo = mfma(a1[s][0], b1v, lo);
    }
    if (more) load_half<HK>(a1, wnext + HALF_WORDS);
    // ---- bias + ReLU, then this lane's share of the second product.  C/D layout of v_mfma_f32_32x32x16_f16: register r of a lane holds row
    // (r & 3) + 8 (r >> 2) + 4 g, i.e. registers 4 q .. 4 q + 3 are 4 CONSECUTIVE channels: b1 and a row of Wf are read 16 bytes at a time.
    const f4* b1l = reinterpret_cast<const f4*>(b1s + l * H + wave * 32 + 4 * g);            // + 2 q float4s: channels 8 q + 4 g ...
    const f4* wfl = reinterpret_cast<const f4*>(wfs + l * H + wave * 32 + 4 * g);            // row o of the level: + o * (n H / 4)
    f4 hv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 bq = b1l[2 * q];
      hv[q] = f4{fmaxf(join(hi[4 * q], lo[4 * q]) + bq.x, 0.f), fmaxf(join(hi[4 * q + 1], lo[4 * q + 1]) + bq.y, 0.f),
                 fmaxf(join(hi[4 * q + 2], lo[4 * q + 2]) + bq.z, 0.f), fmaxf(join(hi[4 * q + 3], lo[4 * q + 3]) + bq.w, 0.f)};
    }
#pragma unroll
    for (int o = 0; o < MAX_OUT; ++o)
      if (o < om) {                                                       // (uniform)
        f4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = wfl[o * (n * H / 4) + 2 * q];
        float acc = part[o];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc = __builtin_fmaf(w[q].x, hv[q].x, acc);
          acc = __builtin_fmaf(w[q].y, hv[q].y, acc);
          acc = __builtin_fmaf(w[q].z, hv[q].z, acc);
          acc = __builtin_fmaf(w[q].w, hv[q].w, acc);
        }
        part[o] = acc;
      }
  }

  // ---- add the two half-waves (channel halves of a pixel), then the waves, in a fixed order
#pragma unroll
  for (int o = 0; o < MAX_OUT; ++o) {
    const float other = __shfl_xor(part[o], 32);
    if (g == 0 && o < om) red[(wave * MAX_OUT + o) * TILE + li] = part[o] + other;
  }
  __syncthreads();
  if (tid < om * TILE) {
    const int o = tid / TILE, p = tid % TILE;
    float acc = bf[br * om + o];
#pragma unroll
    for (int w = 0; w < NW; ++w) acc += red[(w * MAX_OUT + o) * TILE + p];
    if (p0 + p < P) out[((size_t)br * om + o) * P + p0 + p] = acc;
  }
}

constexpr size_t LDS_LIMIT = 160 * 1024;
template <int H>
static size_t lds_bytes(int n, int om) {
  return (size_t)n * (2 * (H / 16) * 2 * TILE * 16) + sizeof(float) * ((size_t)n * H + (size_t)n * om * H + (size_t)(H / 32) * MAX_OUT * TILE);
}

template <int H, int N>
static int launch(const float* feats, const void* w1p, const float* b1, const float* wf, const float* bf, float* out, int P, int om, hipStream_t s) {
  const size_t lds = lds_bytes<H>(N, om);
  if (lds > LDS_LIMIT) return HDN_E_LIMIT;
  static PerDeviceOnce attr;
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_tail_kernel<H, N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr.set(dev_);
  }
  hipLaunchKernelGGL((head_tail_kernel<H, N>), dim3((P + TILE - 1) / TILE, 2), dim3(2 * H), lds, s, feats, static_cast<const u32x4*>(w1p), b1, wf, bf, out, P, om);
  return launch_status();
}

template <int H>
static int launch_levels(const float* feats, const void* w1p, const float* b1, const float* wf, const float* bf, float* out, int n, int P, int om, hipStream_t s) {
  switch (n) {
    case 1: return launch<H, 1>(feats, w1p, b1, wf, bf, out, P, om, s);
    case 2: return launch<H, 2>(feats, w1p, b1, wf, bf, out, P, om, s);
    case 3: return launch<H, 3>(feats, w1p, b1, wf, bf, out, P, om, s);     // MultiBAN / MultiCircBAN: three levels
    case 4: return launch<H, 4>(feats, w1p, b1, wf, bf, out, P, om, s);
    default: return HDN_E_LIMIT;
  }
}

}  // namespace ht
}  // namespace hdn

extern "C" int hdn_head_tail_f32(const float* feats, const void* w1_packed, const float* b1, const float* wf, const float* bf, float* out, int n_levels,
                                 int hidden, int pixels, int n_out, void* stream) {
  if (!feats || !w1_packed || !b1 || !wf || !bf || !out) return HDN_E_NULL;
  if (n_levels <= 0 || hidden <= 0 || pixels <= 0 || n_out <= 0) return HDN_E_SHAPE;
  if (n_levels > 4 || n_out > hdn::ht::MAX_OUT || (hidden != 128 && hidden != 256) || (long long)2 * n_levels * hidden * pixels > 0x7fffffffLL) return HDN_E_LIMIT;
  if (static_cast<const void*>(out) == static_cast<const void*>(feats)) return HDN_E_ALIAS;
  if (!hdn::aligned16(w1_packed)) return HDN_E_LIMIT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (const int rr = hdn::check_fp16_range(feats, (long long)2 * n_levels * hidden * pixels, s)) return rr;
  return hidden == 256 ? hdn::ht::launch_levels<256>(feats, w1_packed, b1, wf, bf, out, n_levels, pixels, n_out, s)
                        : hdn::ht::launch_levels<128>(feats, w1_packed, b1, wf, bf, out, n_levels, pixels, n_out, s);
}
