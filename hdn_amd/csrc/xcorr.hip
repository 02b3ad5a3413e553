// Depthwise cross-correlation kernels (plain and log-polar circular) for gfx950.
//
//   out[p,i,j] = sum_{u,v} xp[p,i+u,j+v] * k[p,u,v]        p = b*C + c  ("plane")
//
// Reference semantics: hdn/core/xcorr.py:37-61.  No channel contraction exists in this
// operator, so there is nothing for MFMA to contract; the production shapes are
// HBM-bound (5 FLOP/B) and the 31x31 (x) 61x61 stress shape is fp32-FMA-bound (82 FLOP/B).
//
// Shape-specialised kernels (compile-time shapes): xcorr_prod29_kernel (5x5 (x) 29x29), xcorr_cfg5_kernel (5x5 (x) 35x35),
// xcorr_north_kernel (31x31 (x) 61x61 direct form; the FFT form is in xcorr_fft.hip),
// xcorr_circ13f_kernel (circular 13x13).  Common structure:
//   * a workgroup's planes are ONE contiguous HBM range for x, k and out, moved with 16-byte coalesced accesses;
//   * taps are wave-uniform and are read through the scalar cache into SGPRs (FMA takes the SGPR operand);
//   * accumulation order is fixed (deterministic); results leave as one contiguous 16-byte coalesced store.
// Circular variant: the padded plane (rows wrap, columns clamp) is never built in HBM.
// (Rounds 1-2 also carried a strip-per-row template "f1" for the two 5x5 shapes and the direct-sum circular kernel
// "circ13r"; both were superseded, had no test of their own, and were removed in round 3.)
//
// Kernel "generic" (runtime shapes): one workgroup per plane, LDS-staged when the plane
// fits, straight from L2 otherwise.  Correct for any Hk<=Hx, Wk<=Wx; not tuned.
#include <atomic>

#include "hdn_common.h"
#include "circ13_tables.h"

#include <type_traits>

#include <cstdlib>

// Measurement hooks (tools/build_variant.sh ... -DHDN_ABLATION -D<experiment>): every site below expands to its production text; the
// experiments' replacement bodies live in ablation/xcorr.inc and are compiled in only under -DHDN_ABLATION, so that editing or adding an
// experiment leaves this translation unit's text (and the hash the committed PMC record carries) unchanged.
#define HDN_ABL_XCORR_0(...) __VA_ARGS__
#define HDN_ABL_XCORR_1(...) __VA_ARGS__
#ifdef HDN_ABLATION
#include "ablation/xcorr.inc"
#endif

namespace hdn {

constexpr int XC_MAX_PROBLEMS = 8;

struct XcorrPtrs {
  const float* x[XC_MAX_PROBLEMS];
  const float* k[XC_MAX_PROBLEMS];
  float* out[XC_MAX_PROBLEMS];
};

// ---------------------------------------------------------------------------------------
// 5x5 (x) 29x29 -> 25x25: the production correlation (3 levels x {cls,loc} per frame, ban.py:76).  HBM-bound.
//
// 4 planes per workgroup = one contiguous HBM range in, one out, one wave per plane, taps in SGPRs, with the LDS access
// pattern made conflict-free: PMC on the first version (horizontal 1x5 strips at row stride 29, lanes 2-way on a bank)
// showed 43 % of its LDS cycles were bank conflicts.  Here a lane owns a
// VERTICAL 5x1 strip (5-row block b, column j; j fastest across lanes) and the plane rows are re-strided to 37
// floats: 25 consecutive columns of one block, then 5*37 = 185 = 25 (mod 32) for the next block => the 32 lanes of
// a bank group always hit 32 distinct banks.  Per tap column a lane reads 9 floats and issues 25 FMAs.
// ---------------------------------------------------------------------------------------
namespace prod29 {
constexpr int HX = 29, WX = 29, HK = 5, WK = 5, HO = 25, WO = 25;
constexpr int XPLANE = HX * WX, OPLANE = HO * WO, KPLANE = HK * WK;
constexpr int SX = 37, LPLANE = HX * SX;   // re-strided LDS plane (1073 floats)
constexpr int PPB = 4;
constexpr int UNITS = 5 * WO;              // 5 row blocks x 25 columns = 125 strips
constexpr int XFLOATS = round_up(PPB * LPLANE + 8, 4);
#ifndef PROD29_ALIAS
#define PROD29_ALIAS 0                     // 1: the outputs are staged OVER the dead inputs (17.2 instead of 27.2 KB of LDS: 8 instead of 5 workgroups per CU), as in
#endif                                     //    xcorr_cfg5_kernel.  Round 2 and round 6 (profiles/round6_experiments.txt section 6) measured it in the step.
#ifndef PROD29_PAD_FLOATS
#define PROD29_PAD_FLOATS 0                // measurement: unused LDS that lowers the number of resident workgroups per CU (5 at 27.2 KB; 3,200: 4; 6,500: 3)
#endif
constexpr int LDS_FLOATS = (PROD29_ALIAS ? XFLOATS : XFLOATS + PPB * OPLANE) + PROD29_PAD_FLOATS;
constexpr int N4 = PPB * XPLANE / 4;       // 841 16-byte loads per workgroup
constexpr int ITER = cdiv(N4, HDN_BLOCK);
}  // namespace prod29

__global__ __launch_bounds__(HDN_BLOCK) void xcorr_prod29_kernel(XcorrPtrs P, int planes) {
  using namespace prod29;
  HDN_ABL_XCORR_0()
  __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
  float* sx = smem;
  float* so = PROD29_ALIAS ? smem : smem + XFLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & (HDN_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prob = blockIdx.y;
  const float* __restrict__ x = P.x[prob];
  const float* __restrict__ k = P.k[prob];
  float* __restrict__ out = P.out[prob];
  const int plane0 = blockIdx.x * PPB;
  const int np = min(PPB, planes - plane0);
  const float* xg = x + size_t(plane0) * XPLANE;

  // ---- stage: all 16-byte loads in flight, then scatter into the re-strided image ------------------------
  if (np == PPB && aligned16(xg)) {
    const float4* s4 = reinterpret_cast<const float4*>(xg);
    float4 r[ITER];
#pragma unroll
    for (int q = 0; q < ITER; ++q) r[q] = ld_stream(s4 + min(tid + q * HDN_BLOCK, N4 - 1));
#pragma unroll
    for (int q = 0; q < ITER; ++q) {
      const int i4 = tid + q * HDN_BLOCK;
      if (i4 < N4) {
        const float v[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
        // element e of the group lives at e + (SX - WX) * (e / WX): the planes of a group are consecutive rows and
        // LPLANE = HX * SX, so one division per 16-byte chunk and a carry per element do the whole re-striding
        const int e0 = 4 * i4, row0 = e0 / WX, c0 = e0 - row0 * WX;
        if (c0 + 3 < WX) {  // the four floats stay in one row: one aligned 16-byte store (e0 and (SX - WX) * row0 are multiples of 4)
          *reinterpret_cast<float4*>(sx + e0 + (SX - WX) * row0) = r[q];
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) sx[e0 + t + (SX - WX) * (row0 + (c0 + t >= WX ? 1 : 0))] = v[t];
        }
      }
    }
  } else {
    for (int e = tid; e < np * XPLANE; e += HDN_BLOCK) {
      const int p = e / XPLANE, rem = e - p * XPLANE;
      const int rr = rem / WX, c = rem - rr * WX;
      sx[p * LPLANE + rr * SX + c] = xg[e];
    }
  }
  __syncthreads();

  // ---- correlate: one wave per plane, one lane per 5x1 output strip --------------------------------------
  float held[2][5];     // (PROD29_ALIAS: the strips wait in registers until every wave is done reading the image)
  if (wave < np) {  // wave-uniform
    const float* __restrict__ kp = k + size_t(plane0 + wave) * KPLANE;  // wave-uniform -> scalar loads
    const float* xs = sx + wave * LPLANE;
    float* os = so + wave * OPLANE;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int unit = rd * HDN_WAVE + lane;
      const int uu = min(unit, UNITS - 1);
      const int b = uu / WO, j = uu - b * WO;
      const float* xc = xs + (5 * b) * SX + j;
      float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < WK; ++v) {
        float col[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) col[r] = xc[r * SX + v];
#pragma unroll
        for (int u = 0; u < HK; ++u) {
          const float kv = kp[u * WK + v];
#pragma unroll
          for (int t = 0; t < 5; ++t) acc[t] = __builtin_fmaf(col[t + u], kv, acc[t]);
        }
      }
      if constexpr (PROD29_ALIAS) {
#pragma unroll
        for (int t = 0; t < 5; ++t) held[rd][t] = acc[t];
      } else if (unit < UNITS) {
#pragma unroll
        for (int t = 0; t < 5; ++t) os[(5 * b + t) * WO + j] = acc[t];
      }
    }
  }
  __syncthreads();
  if constexpr (PROD29_ALIAS) {   // nobody reads the inputs any more
    if (wave < np) {
      float* os = so + wave * OPLANE;
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        const int unit = rd * HDN_WAVE + lane;
        if (unit < UNITS) {
          const int b = unit / WO, j = unit - b * WO;
#pragma unroll
          for (int t = 0; t < 5; ++t) os[(5 * b + t) * WO + j] = held[rd][t];
        }
      }
    }
    __syncthreads();
  }

  float* og = out + size_t(plane0) * OPLANE;
  if (np == PPB && aligned16(og)) copy_l2g_full<PPB * OPLANE>(so, og, tid);
  else copy_l2g(so, og, np * OPLANE, tid);
}

// ---------------------------------------------------------------------------------------
// 5x5 (x) 35x35 -> 31x31: the production correlation at TRACK.INSTANCE_SIZE 303 (BASELINE config 5: search features 37x37
// -> conv_search 35x35, score map 31x31, hdn_tracker_proj_e2e.py:24-25).  HBM-bound like the 29x29 one.
//
// Same workgroup structure as xcorr_prod29_kernel (4 planes = one contiguous 16-byte aligned HBM range in and out, one
// wave per plane, the 25 taps in SGPRs, results staged in LDS).  The output is 31 wide, so a 32-lane LDS bank group is
// given exactly ONE block of 31 columns (lane 31 of the group repeats column 30: same address, a broadcast): whatever
// the row stride, the 32 lanes of a group then read 31 consecutive floats, i.e. conflict-free on the LINEAR image, and
// the re-striding scatter of the 29x29 kernel is not needed.  A lane owns a VERTICAL 8x1 strip: per tap column it reads
// 12 floats (6 ds_read2_b32 = 6 vertical operand pairs) and issues 22 v_pk_fma_f32 (cfg5::strip below); 4 row blocks x 31
// columns = 124 strips = two rounds of a wave (97 % of the lanes), the 32nd row of the last block is computed and dropped (its
// taps read one row beyond the plane: the next plane's first row or the pad behind the last plane).
// Round 3 (profiles/round3_cfg5_experiments.txt): the kernel runs at the rate the memory system gives its traffic SHAPE
// (19.6 KB in, 15.4 KB out per workgroup, tools/experiments/ubench_stream.hip: 157.9 us with no arithmetic at all at 8 workgroups
// per CU; this kernel 157-159 us).  Fewer resident workgroups would stream faster (144.5 us at 4 per CU) but then no longer hide
// each other's barriers and LDS phases (176 us); a persistent form with the next group fetched by LDS-DMA into a second image
// (4 per CU, a group in flight per workgroup all the time) was built, bit-identical, and measured at 160 us: removed again.
// ---------------------------------------------------------------------------------------
namespace cfg5 {
constexpr int HX = 35, WX = 35, HK = 5, WK = 5, HO = 31, WO = 31;
constexpr int XPLANE = HX * WX, OPLANE = HO * WO, KPLANE = HK * WK;
constexpr int PPB = 4, TH = 8;                       // planes per workgroup, strip height
constexpr int XFLOATS = round_up(PPB * XPLANE + WX + 4, 4);   // + the row the last block's dropped output reads
constexpr int LDS_FLOATS = XFLOATS + PPB * OPLANE;   // 8,784 floats = 35 KB: 4 workgroups per CU

// One 8 x 1 output strip (rows r0 .. r0 + 7 of column j) of one plane, packed over VERTICAL neighbours.
//   P[m] = (x[r0 + 2m][c], x[r0 + 2m + 1][c])   one ds_read2_b32 each, 6 per tap column c = j + v (12 input rows)
//   even tap rows u = 2w:      accE[q] = (out[2q],     out[2q + 1]) += P[q + w] * k[u][v]      q = 0..3
//   odd  tap rows u = 2w + 1:  accO[q] = (out[2q - 1], out[2q])     += P[q + w] * k[u][v]      q = 0..4
// (the operand pair of an odd tap row is aligned again once the OUTPUT pair is shifted by one row: the scheme of the 31x31
// direct kernel, turned by 90 degrees), out[2q] = accE[q].x + accO[q].y, out[2q + 1] = accE[q].y + accO[q + 1].x.
// 110 v_pk_fma_f32 + 8 adds and 30 LDS reads per strip; the compiler's own packing of the scalar form needed 100 v_pk_fma_f32
// plus ~107 v_mov_b32 to build its operand pairs.  `a` = LDS byte address of x[r0][j].
typedef const float __attribute__((address_space(4))) cfloat;
template <int V, int M>
__device__ __forceinline__ void strip_pairs(float2v (&Pm)[6], uint32_t a0, uint32_t a1) {
  if constexpr (M < 6) {
    if constexpr (M < 3) Pm[M] = lds_read_pair<(2 * M) * WX + V, (2 * M + 1) * WX + V>(a0);
    else Pm[M] = lds_read_pair<(2 * M - 6) * WX + V, (2 * M - 5) * WX + V>(a1);
    strip_pairs<V, M + 1>(Pm, a0, a1);
  }
}
template <int V>
__device__ __forceinline__ void strip_column(float2v (&accE)[4], float2v (&accO)[5], uint32_t a0, uint32_t a1, const cfloat* kp) {
  float2v Pm[6];
  strip_pairs<V, 0>(Pm, a0, a1);
  lds_wait_all();
#pragma unroll
  for (int m = 0; m < 6; ++m) pin(Pm[m]);
#pragma unroll
  for (int u = 0; u < HK; ++u) {
    const float kv = kp[u * WK + V];
    const float2v kk = {kv, kv};
    if (u % 2 == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) accE[q] = __builtin_elementwise_fma(Pm[q + u / 2], kk, accE[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 5; ++q) accO[q] = __builtin_elementwise_fma(Pm[q + u / 2], kk, accO[q]);
    }
  }
}
__device__ __forceinline__ void strip(float (&acc)[TH], const float* x00, const cfloat* kp) {
  float2v accE[4], accO[5];
#pragma unroll
  for (int q = 0; q < 4; ++q) accE[q] = float2v{0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 5; ++q) accO[q] = float2v{0.f, 0.f};
  const uint32_t a0 = lds_addr(x00), a1 = a0 + 6 * WX * 4;
  strip_column<0>(accE, accO, a0, a1, kp);
  strip_column<1>(accE, accO, a0, a1, kp);
  strip_column<2>(accE, accO, a0, a1, kp);
  strip_column<3>(accE, accO, a0, a1, kp);
  strip_column<4>(accE, accO, a0, a1, kp);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc[2 * q] = accE[q].x + accO[q].y;
    acc[2 * q + 1] = accE[q].y + accO[q + 1].x;
  }
}
__device__ __forceinline__ const cfloat* scalar_ptr(const float* p) {  // wave-uniform pointer -> constant address space: scalar loads
  uint64_t a = reinterpret_cast<uint64_t>(p);
  asm volatile("" : "+s"(a));
  return (const cfloat*)a;
}
}  // namespace cfg5

__global__ __launch_bounds__(HDN_BLOCK) void xcorr_cfg5_kernel(XcorrPtrs P, int planes) {
  using namespace cfg5;
  __shared__ __attribute__((aligned(16))) float smem[XFLOATS];
  float* sx = smem;
  float* so = smem;  // the outputs are staged OVER the inputs once every wave is done reading (19.8 KB: 8 workgroups per CU)

  const int tid = threadIdx.x;
  const int lane = tid & (HDN_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prob = blockIdx.y;
  const float* __restrict__ x = P.x[prob];
  const float* __restrict__ k = P.k[prob];
  float* __restrict__ out = P.out[prob];
  const int plane0 = blockIdx.x * PPB;
  const int np = min(PPB, planes - plane0);
  const float* xg = x + size_t(plane0) * XPLANE;

  if (np == PPB && aligned16(xg)) copy_g2l_full<PPB * XPLANE>(xg, sx, tid);  // all 16-byte loads in flight at once
  else copy_g2l(xg, sx, np * XPLANE, tid);
  if (tid < WX + 4) sx[PPB * XPLANE + tid] = 0.f;  // the pad row (read by dropped outputs only; kept finite)
  __syncthreads();

  float accs[2][TH];
  if (wave < np) {  // wave-uniform
    const cfloat* kp = scalar_ptr(k + size_t(plane0 + wave) * KPLANE);  // wave-uniform -> scalar loads
    const float* xs = sx + wave * XPLANE;
    const int j = min(lane & 31, WO - 1);
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int b = 2 * rd + (lane >> 5);  // one row block per 32-lane bank group
      strip(accs[rd], xs + (TH * b) * WX + j, kp);
    }
  }
  __syncthreads();  // nobody reads the inputs any more
  if (wave < np && (lane & 31) < WO) {
    float* os = so + wave * OPLANE;
    const int j = lane & 31;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int b = 2 * rd + (lane >> 5);
#pragma unroll
      for (int t = 0; t < TH; ++t)
        if (TH * b + t < HO) os[(TH * b + t) * WO + j] = accs[rd][t];
    }
  }
  __syncthreads();

  float* og = out + size_t(plane0) * OPLANE;
  if (np == PPB && aligned16(og)) copy_l2g_full<PPB * OPLANE>(so, og, tid);
  else copy_l2g(so, og, np * OPLANE, tid);
}

// ---------------------------------------------------------------------------------------
// 31x31 (x) 61x61 -> 31x31 (BASELINE.json north-star shape).  fp32-FMA-bound: 1.85 MFLOP per 23 KB plane.
//
// gfx950 only reaches its fp32 vector peak through v_pk_fma_f32 (measured 151 TF vs 75 TF for v_fma_f32,
// profiles/round1_ubench_fma.txt), and at the 2 waves/SIMD this kernel's LDS footprint allows only 16-byte
// LDS reads run at full rate.  Both are met with adjacent-column operand pairs R[m] = (x[2m], x[2m+1]):
//   * even taps v = 2w update even-aligned output pairs  accE[j] = (out[2j],   out[2j+1]) += R[j+w] * k[v]
//   * odd  taps v = 2w+1 update ODD-aligned output pairs accO[j] = (out[2j-1], out[2j])   += R[j+w] * k[v]
//     (the operand pair of an odd tap is aligned again once the OUTPUT pair is shifted by one column),
//   and out[2j] = accE[j].lo + accO[j].hi, out[2j+1] = accE[j].hi + accO[j+1].lo at the end of the plane.
//   No register shuffles, every LDS read is a 16-byte ds_read_b128, 16 independent accumulator chains.
//   * a lane owns 16 outputs of one row (2 lanes per row, 62 per plane); a tap row costs 12 ds_read_b128 and
//     16*8 + 15*8 = 248 packed FMAs, the minimum: the odd-aligned pair straddling the two lanes of a row is
//     accumulated by the right-hand lane only and handed over with one cross-lane read per plane;
//   * the 31 taps of a kernel row are wave-uniform: s_load into SGPRs, broadcast by op_sel;
//   * tap row u+1 (LDS + SGPRs) is fetched while row u's FMAs issue.
// Waves are autonomous and persistent: a wave owns an LDS slot (rows re-strided to 68 floats so the b128 reads
// of 16 consecutive rows hit 64 distinct banks) and streams planes through it; the NEXT plane's 15 x 16-byte
// global loads are in flight in registers while the current plane is correlated, and results go straight to HBM.
// There is no workgroup barrier and no bulk fill/compute/store phase.
// LDS reads, LDS writes and scalar tap loads are inline asm (ordered among themselves, waited for by hand:
// cdna guide §5.7); the compiler only sees global loads/stores, address arithmetic and the FMAs.
// ---------------------------------------------------------------------------------------
namespace north {
constexpr int HX = 61, WX = 61, HK = 31, WK = 31, HO = 31, WO = 31;
constexpr int XPLANE = HX * WX;         // 3721
constexpr int OPLANE = HO * WO;         // 961
constexpr int SX = 68;                  // LDS row stride (floats): 16-byte aligned, 68 = 4 mod 64
constexpr int NQUAD = 12;               // ds_read_b128 per tap row: x[16s .. 16s+47]
constexpr int NQ = 15;                  // 16-byte global loads per lane per plane: 64*15*4 = 3840 >= 3721 + 3
constexpr int WINDOW = NQ * HDN_WAVE * 4;
constexpr int GUARD_ROWS_BEFORE = 1, GUARD_ROWS_AFTER = 3;  // window floats outside the plane land here
constexpr int SLOT = (GUARD_ROWS_BEFORE + HX + GUARD_ROWS_AFTER) * SX;  // 4420 floats per wave
constexpr size_t LDS_BYTES = size_t(4 * SLOT) * sizeof(float);          // 70,720 B: two workgroups per CU

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

struct Taps {  // one kernel row in SGPRs: k[0..15], k[16..23], k[24..27], k[28..29], k[30]
  f16v a;
  f8v b;
  f4v c;
  float2v d;
  float e;
  __device__ __forceinline__ float get(int v) const {
    return v < 16 ? a[v] : (v < 24 ? b[v - 16] : (v < 28 ? c[v - 24] : (v < 30 ? d[v - 28] : e)));
  }
};

struct Row {
  f4v Q[NQUAD];  // Q[m] = x[4m .. 4m+3] of the lane's 48-float window; pair R[n] = (x[2n], x[2n+1])
  Taps k;
  __device__ __forceinline__ float2v pair(int n) const { return (n & 1) ? Q[n >> 1].zw : Q[n >> 1].xy; }
};

template <int M>
__device__ __forceinline__ void issue_quads(Row& R, uint32_t a) {
  if constexpr (M < NQUAD) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(R.Q[M]) : "v"(a), "n"(M * 16));
    issue_quads<M + 1>(R, a);
  }
}

// Issue the 12 LDS reads and the 31 scalar tap loads of one kernel row.  Nothing is waited for here.
__device__ __forceinline__ void load_row(Row& R, uint32_t xaddr, const float* kr) {
  issue_quads<0>(R, xaddr);
  asm volatile(
      "s_nop 4\n\t"
      "s_load_dwordx16 %0, %5, 0x0\n\t"
      "s_load_dwordx8 %1, %5, 0x40\n\t"
      "s_load_dwordx4 %2, %5, 0x60\n\t"
      "s_load_dwordx2 %3, %5, 0x70\n\t"
      "s_load_dword %4, %5, 0x78"
      : "=&s"(R.k.a), "=&s"(R.k.b), "=&s"(R.k.c), "=&s"(R.k.d), "=&s"(R.k.e)
      : "s"(kr));
}

// All outstanding LDS reads and scalar loads have landed; pin their destinations so no FMA floats above the wait.
__device__ __forceinline__ void land_row(Row& R) {
  lds_wait_all();
#pragma unroll
  for (int m = 0; m < NQUAD; ++m) asm volatile("" : "+v"(R.Q[m]));
  asm volatile("" : "+s"(R.k.a), "+s"(R.k.b), "+s"(R.k.c), "+s"(R.k.d), "+s"(R.k.e));
  __builtin_amdgcn_sched_barrier(0);
}

// A tap that is exactly +-0 contributes nothing (for finite x) and its 8 packed FMAs are skipped by a SCALAR branch:
// taps live in SGPRs, so the test is wave-uniform and costs no divergence.  The correlation kernels are the outputs of
// conv+BN+ReLU (ban.py:55-58): about half of their taps are exact zeros (50 % in the synthetic relu(N(0,1)) inputs
// of BASELINE config 2).  Non-finite x under a zero tap is the only case where the result differs (NaN in the reference).
// The FMAs are inline asm with the accumulator as a read-write operand: the update is in place by construction, so the
// branches introduce no register copies at their join points (the C++ form got 282 v_mov_b64 from the compiler).
__device__ __forceinline__ bool tap_nonzero(float kv) {
  unsigned b;
  asm("s_and_b32 %0, %1, 0x7fffffff" : "=s"(b) : "s"(kv));  // integer test on the SGPR: s_cmp + s_cbranch_scc
  return b != 0u;
}

// acc += P * k.lo (LO) or P * k.hi (!LO), both halves of the result using the same tap (op_sel broadcast)
template <bool LO>
__device__ __forceinline__ void pk_fma_tap(float2v& acc, const float2v& P, const float2v& kpair) {
  if constexpr (LO) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(P), "s"(kpair));
  else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(P), "s"(kpair));
}

template <bool SKIP_ZERO_TAPS>
__device__ __forceinline__ void fma_row(float2v (&accE)[8], float2v (&accO)[8], const Row& R) {
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    // taps 2w (even) and 2w+1 (odd) share the SGPR pair (k[2w], k[2w+1]); tap 30 has no partner
    const float2v kp = {R.k.get(2 * w), w < 15 ? R.k.get(2 * w + 1) : 0.f};
    if (!SKIP_ZERO_TAPS || tap_nonzero(kp.x)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pk_fma_tap<true>(accE[j], R.pair(j + w), kp);
    }
    if (w < 15) {
      if (!SKIP_ZERO_TAPS || tap_nonzero(kp.y)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pk_fma_tap<false>(accO[j], R.pair(j + w), kp);
      }
    }
  }
}

// The 16-byte-aligned window [first, first + 3840) floats that contains a plane (first = plane start rounded down
// to 16 B).  Lane l holds float4 #(l + 64 q).  Windows that would leave the tensor use guarded scalar loads.
__device__ __forceinline__ void fetch_plane(f4v (&R)[NQ], const float* __restrict__ x, long long first, long long total,
                                            int lane) {
  const bool inside = first >= 0 && first + WINDOW <= total;  // wave-uniform
  if (inside) {
    const f4v* w = reinterpret_cast<const f4v*>(x + first);
#pragma unroll
    for (int q = 0; q < NQ; ++q) R[q] = w[lane + q * HDN_WAVE];
  } else {
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
      const long long f = first + 4 * (lane + q * HDN_WAVE);
      f4v v;
      v.x = (f + 0 >= 0 && f + 0 < total) ? x[f + 0] : 0.f;
      v.y = (f + 1 >= 0 && f + 1 < total) ? x[f + 1] : 0.f;
      v.z = (f + 2 >= 0 && f + 2 < total) ? x[f + 2] : 0.f;
      v.w = (f + 3 >= 0 && f + 3 < total) ? x[f + 3] : 0.f;
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq)
        if (qq == q) R[qq] = v;  // static register indices
    }
  }
}

// Scatter the window into the wave's slot with rows re-strided to SX.  Window float f is plane element f - shift;
// elements outside [0, 3721) fall into the guard rows.  slot_a = byte address of the slot.
__device__ __forceinline__ void stash_plane(const f4v (&R)[NQ], uint32_t slot_a, int shift, int lane) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = 4 * (lane + q * HDN_WAVE) - shift + WX;  // >= 58: one guard row of bias keeps the division unsigned
    const int row = e / WX, col = e - row * WX;            // row 0 = guard row before the plane
    const uint32_t a = slot_a + uint32_t(row * SX + col) * 4u;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // element t sits at column col + t, or wraps into the next row (+SX - WX floats further on)
      const uint32_t at = a + 4u * t + ((col + t >= WX) ? uint32_t(SX - WX) * 4u : 0u);
      asm volatile("ds_write_b32 %0, %1" : : "v"(at), "v"(R[q][t]));
    }
  }
}
}  // namespace north

// One plane: 31 tap rows, row u+1's operands in flight while row u's FMAs issue.
template <bool SKIP_ZERO_TAPS>
__device__ __forceinline__ void north_plane(float2v (&accE)[8], float2v (&accO)[8], uint32_t xa, const float* kp) {
  using namespace north;
  Row A, B;
  load_row(A, xa, kp);
  land_row(A);
#pragma unroll 1
  for (int u = 0; u < HK - 1; u += 2) {
    load_row(B, xa + (u + 1) * (SX * 4), kp + (u + 1) * WK);
    __builtin_amdgcn_sched_barrier(0);
    fma_row<SKIP_ZERO_TAPS>(accE, accO, A);
    land_row(B);
    load_row(A, xa + (u + 2) * (SX * 4), kp + (u + 2) * WK);
    __builtin_amdgcn_sched_barrier(0);
    fma_row<SKIP_ZERO_TAPS>(accE, accO, B);
    land_row(A);
  }
  fma_row<SKIP_ZERO_TAPS>(accE, accO, A);  // u = 30
}

// MODE 1 (the one instantiated): exact-zero taps are skipped; MODE 0: dense FMA stream (rounds 1-3, 326-352 us).  A skipped tap costs a
// taken scalar branch, ~20 clocks against 32 for its 8 packed FMAs; a kept one ~4 extra: skipping wins above ~25 % zero
// taps (post-ReLU kernels: ~50 %), loses 12 % on fully dense taps.  A per-plane choice between both streams inside one
// kernel was measured slower than either (register spills + two unrolled streams): profiles/round1_north_zero_taps.txt.
template <int MODE>
__global__ __launch_bounds__(HDN_BLOCK, 2) void xcorr_north_kernel(XcorrPtrs P, int planes) {
  using namespace north;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int lane = threadIdx.x & (HDN_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int prob = blockIdx.y;
  const float* __restrict__ x = P.x[prob];
  const float* __restrict__ k = P.k[prob];
  float* __restrict__ out = P.out[prob];
  const int gw = blockIdx.x * 4 + wave;  // this wave's id; it owns planes gw, gw + nw, ...
  const int nw = gridDim.x * 4;
  if (gw >= planes) return;

  const uint32_t slot_a = lds_addr(smem + wave * SLOT);
  const uint32_t plane_a = slot_a + uint32_t(GUARD_ROWS_BEFORE * SX) * 4u;  // LDS address of plane element (0,0)
  const long long total = (long long)planes * XPLANE;
  // x may be any 4-byte aligned pointer: mis = float offset of x inside its 16-byte line
  const int mis = int((reinterpret_cast<uintptr_t>(x) >> 2) & 3);

  const int i = min(lane & 31, HO - 1);  // output row (lanes 31 and 63 shadow row 30 and do not store)
  const int s = lane >> 5;               // output columns [16 s, 16 s + 16)
  const bool live = (lane & 31) < HO;
  const uint32_t xa = plane_a + uint32_t(i * SX + s * 16) * 4u;

  f4v R[NQ];
  {
    const long long start = (long long)gw * XPLANE;
    fetch_plane(R, x, start - ((start + mis) & 3), total, lane);
  }
#pragma unroll 1
  for (int plane = gw; plane < planes; plane += nw) {
    const long long start = (long long)plane * XPLANE;
    stash_plane(R, slot_a, int((start + mis) & 3), lane);
    if (plane + nw < planes) {  // wave-uniform: next plane's loads fly during this plane's FMAs
      const long long nstart = (long long)(plane + nw) * XPLANE;
      fetch_plane(R, x, nstart - ((nstart + mis) & 3), total, lane);
    }
    const float* kp = k + size_t(plane) * (HK * WK);
    float2v accE[8], accO[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) accE[j] = accO[j] = float2v{0.f, 0.f};
    north_plane<MODE == 1>(accE, accO, xa, kp);
    // accO[0] of the right-hand lane (s = 1) is (col 15, col 16): its low half is the odd-tap sum of the left-hand
    // lane's last column, fetched across the half-waves once per plane instead of accumulating a 9th pair per row.
    const float odd15 = __shfl(accO[0].x, (lane & 31) + 32, HDN_WAVE);
    if (live) {
      float* o = out + size_t(plane) * OPLANE + i * WO + s * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[2 * j] = accE[j].x + accO[j].y;
        if (j < 7) o[2 * j + 1] = accE[j].y + accO[j + 1].x;
      }
      if (s == 0) o[15] = accE[7].y + odd15;  // column 31 (s = 1) does not exist
    }
  }
}

// ---------------------------------------------------------------------------------------
// 13x13 circular, DFT form (default since round 2).  The row axis of the padded plane wraps, so a 13-point DFT along it turns
// the 13 x 13 tap sum of every output into 7 independent complex 1-D correlations along the (clamped) column axis:
//   X_f[s] = sum_r x[r][s] w^(f r),  K_f[j] = sum_u k[u][j] w^(f u),  w = exp(-2 pi i / 13), f = 0..6 (real input)
//   Z_f[c] = sum_j conj(K_f[j]) X_f[clamp(c + j - 6)]                 (taps that meet a replicated column are summed first)
//   out[i][c] = 1/13 (Z_0[c] + 2 sum_{f>0} Re(Z_f[c] w^(-f (i + 7))))
// about 3 k real multiply-adds per plane instead of 28.6 k (the direct kernel above executes 24 k).
// A WAVE owns 9 consecutive planes and never meets a barrier; its LDS slot holds them at a stride of 172 floats.
//   stage 1  lane = (plane, column): the 13-point DFT of its column of x and of k as 6 sums / differences of rows r, 13 - r
//            times the 6 base twiddles (SGPR operands), 117 columns in two rounds; spectra to LDS as [plane][f][column]
//   stage 2  lane = (plane, f), 63 lanes: the clamped correlation along the columns; Z to LDS as [plane][column][f]
//   stage 3  lane = (plane, column): the inverse for the 13 rows of its column, rows m, 13 - m together; rows staged in LDS and
//            written out 16 bytes per lane.
// Accuracy: fp32 throughout; against float64 the result is closer than the direct fp32 sum (tests: same bounds).
// ---------------------------------------------------------------------------------------
namespace circ13f {
constexpr int N = 13, PL = N * N, NF = 7, PPW = 9, WAVES = HDN_BLOCK / 64;
constexpr int XS = 172;                    // floats per plane in the LDS input / output images (43 16-byte words)
// floats per plane in the spectra and Z images: 7 x 13 complex = 91 float2, NOT padded: 91 = 27 (mod 32) is the plane pitch that
// spreads the 8-byte slots of a 32-lane access best for the (plane, frequency) and (plane, column) lane orders of stages 2 and 3
// (brute force over pitches and row strides, round 3: 175 LDS cycles per wave against 257 at the former pitch of 92; SQ_LDS_BANK_CONFLICT
// was 45 % of the kernel's LDS cycles)
constexpr int TS = 182;
constexpr int REGION = round_up(PPW * TS, 4);   // 1,640 floats: a region holds 9 input planes, or 9 spectra, or Z, or the output rows (16-byte aligned)
constexpr int WAVE_FLOATS = 2 * REGION;    // 13.2 KB per wave: 12 waves per CU
static_assert(PPW * XS <= REGION, "inputs fit");

typedef float f4e __attribute__((ext_vector_type(4)));
typedef f4e f4u __attribute__((aligned(4)));   // a 16-byte access that is only 4-byte aligned (HBM side: plane = 676 bytes)

__device__ __forceinline__ float2v cmac_conj(float2v acc, float2v k, float2v x) {  // acc += conj(k) * x
  acc = __builtin_elementwise_fma(x, k.xx, acc);
  return __builtin_elementwise_fma(float2v{x.y, -x.x}, k.yy, acc);
}

// the 6 base twiddles (cos, sin)(2 pi k / 13) and their conjugates, wave-uniform (SGPRs)
struct Tw {
  float2v p[6], m[6];  // p[k-1] = (cos, sin), m[k-1] = (cos, -sin)
};
// (cos, -sin)(2 pi n / 13) for any n >= 0: the forward twiddle w^n
__device__ __forceinline__ float2v tw_fwd(const Tw& t, int n) {
  const int k = n % N;
  return k <= 6 ? t.m[k - 1] : t.p[N - k - 1];
}
// (cos, sin)(2 pi n / 13)
__device__ __forceinline__ float2v tw_inv(const Tw& t, int n) {
  const int k = n % N;
  return k <= 6 ? t.p[k - 1] : t.m[N - k - 1];
}

// stage 1 for one tensor: column s of plane q -> its 7 spectral values S[f] (S[0] = (sum, 0))
__device__ __forceinline__ void dft_col(float2v (&S)[NF], const float* col, const Tw& t) {
  float v[N];
#pragma unroll
  for (int r = 0; r < N; ++r) v[r] = col[r * N];
  float2v sd[7];  // (v[r] + v[13 - r], v[r] - v[13 - r]), r = 1..6
  float s0 = v[0];
#pragma unroll
  for (int r = 1; r <= 6; ++r) {
    sd[r] = float2v{v[r] + v[N - r], v[r] - v[N - r]};
    s0 += sd[r].x;
  }
  S[0] = float2v{s0, 0.f};
#pragma unroll
  for (int f = 1; f < NF; ++f) {
    float2v a = float2v{v[0], 0.f};
#pragma unroll
    for (int r = 1; r <= 6; ++r) a = __builtin_elementwise_fma(sd[r], tw_fwd(t, f * r), a);
    S[f] = a;
  }
}

struct Moves {
  static constexpr int WPP = 42, ROUNDS = (PPW * WPP + 63) / 64;  // 42 16-byte words per plane (+ one float): 6 rounds of 64 lanes
  int g[ROUNDS], l[ROUNDS];                                       // float offsets of the word in HBM (-1: no task) / in the LDS image
};
__device__ __forceinline__ Moves make_moves(int np, int lane) {
  Moves m;
#pragma unroll
  for (int rd = 0; rd < Moves::ROUNDS; ++rd) {
    const int t = lane + 64 * rd;
    const int q = t / Moves::WPP, w = t - q * Moves::WPP;
    m.g[rd] = q < np ? q * PL + 4 * w : -1;
    m.l[rd] = q * XS + 4 * w;
  }
  return m;
}
struct PlaneRegs {
  f4e v[Moves::ROUNDS];
  float last;
};
__device__ __forceinline__ void planes_load(PlaneRegs& r, const float* __restrict__ g, const Moves& m, int np, int lane) {
#pragma unroll
  // all in flight at once.  (No streaming hint here: with nontemporal loads / stores this kernel is 8 % SLOWER, 42-45 vs 38-41 us -
  // its 16-byte accesses are only 4-byte aligned - while the 5x5 kernels and the 31x31 kernel gain 3-9 %.)
  for (int rd = 0; rd < Moves::ROUNDS; ++rd) r.v[rd] = *reinterpret_cast<const f4u*>(g + max(m.g[rd], 0));
  r.last = g[min(lane, np - 1) * PL + 168];
}
__device__ __forceinline__ void planes_store(float* lds, const PlaneRegs& r, const Moves& m, int np, int lane) {
#pragma unroll
  for (int rd = 0; rd < Moves::ROUNDS; ++rd)
    if (m.g[rd] >= 0) *reinterpret_cast<f4e*>(lds + m.l[rd]) = r.v[rd];
  if (lane < np) lds[lane * XS + 168] = r.last;
}
__device__ __forceinline__ void planes_out(const float* lds, float* __restrict__ g, const Moves& m, int np, int lane) {
#pragma unroll
  for (int rd = 0; rd < Moves::ROUNDS; ++rd)
    if (m.g[rd] >= 0) *reinterpret_cast<f4u*>(g + m.g[rd]) = *reinterpret_cast<const f4e*>(lds + m.l[rd]);
  if (lane < np) g[lane * PL + 168] = lds[lane * XS + 168];
}
}  // namespace circ13f

__global__ __launch_bounds__(HDN_BLOCK, 3) void xcorr_circ13f_kernel(XcorrPtrs P, int planes, int groups_per_problem, int total_groups) {
  using namespace circ13f;
  HDN_ABL_XCORR_1()
#ifndef CIRC13_PAD_FLOATS
#define CIRC13_PAD_FLOATS 0      // measurement: unused LDS that lowers the resident workgroups per CU from 3 (6,600: 2)
#endif
  __shared__ __attribute__((aligned(16))) float smem[WAVES * WAVE_FLOATS + CIRC13_PAD_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.x * WAVES + wave;
  if (g >= total_groups) return;
  const int prob = g / groups_per_problem, p0 = (g - prob * groups_per_problem) * PPW, np = min(PPW, planes - p0);
  float* ra = smem + wave * WAVE_FLOATS;  // region A: x image -> x spectra -> Z
  float* rb = ra + REGION;                // region B: k image -> k spectra -> output rows
  const Moves mv = make_moves(np, lane);
  PlaneRegs rx, rk;
  planes_load(rx, P.x[prob] + size_t(p0) * PL, mv, np, lane);
  planes_load(rk, P.k[prob] + size_t(p0) * PL, mv, np, lane);
  Tw tw;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    tw.p[k] = float2v{CIRC13_CS[2 * k], CIRC13_CS[2 * k + 1]};
    tw.m[k] = float2v{CIRC13_CS[2 * k], -CIRC13_CS[2 * k + 1]};
  }
  planes_store(ra, rx, mv, np, lane);
  planes_store(rb, rk, mv, np, lane);

  // ---- stage 1: lane = (plane, column), two rounds; both rounds read before the spectra go over the images ----
  float2v SX[2][NF], SK[2][NF];
  int sq[2], ss[2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int t = min(lane + 64 * rd, np * N - 1);
    sq[rd] = t / N;
    ss[rd] = t - sq[rd] * N;
    dft_col(SX[rd], ra + sq[rd] * XS + ss[rd], tw);
    dft_col(SK[rd], rb + sq[rd] * XS + ss[rd], tw);
  }
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    if (lane + 64 * rd < np * N) {
      float2v* xp = reinterpret_cast<float2v*>(ra + sq[rd] * TS) + ss[rd];
      float2v* kp = reinterpret_cast<float2v*>(rb + sq[rd] * TS) + ss[rd];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        xp[f * N] = SX[rd][f];
        kp[f * N] = SK[rd][f];
      }
    }
  }

  // ---- stage 2: lane = (plane q, frequency f) ----
  {
    const int q = lane / NF, f = lane - q * NF;
    const bool live = q < np;  // lane 63 and the planes a short last group lacks redo plane 0: in bounds, unused
    const int qa = live ? q : 0;
    const float2v* xp = reinterpret_cast<const float2v*>(ra + qa * TS) + f * N;
    const float2v* kp = reinterpret_cast<const float2v*>(rb + qa * TS) + f * N;
    float2v X[N], K[N];
#pragma unroll
    for (int s = 0; s < N; ++s) {
      X[s] = xp[s];
      K[s] = kp[s];
    }
    // prefix / suffix sums of K over the taps that meet the replicated first / last column
    float2v PS[7], SS[7];  // PS[m] = K[0] + .. + K[m];  SS[m] = K[12 - m] + .. + K[12]
    PS[0] = K[0];
    SS[0] = K[12];
#pragma unroll
    for (int m = 1; m < 7; ++m) {
      PS[m] = PS[m - 1] + K[m];
      SS[m] = SS[m - 1] + K[12 - m];
    }
    const float zs = f == 0 ? 1.f / 13.f : 2.f / 13.f;  // the inverse's weights: the conjugate half of the spectrum counts twice
    float2v Z[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
      float2v z = float2v{0.f, 0.f};
      if (c <= 6) z = cmac_conj(z, PS[6 - c], X[0]);    // taps j <= 6 - c read column 0
      if (c >= 6) z = cmac_conj(z, SS[c - 6], X[12]);   // taps j >= 18 - c read column 12
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int s = c + j - 6;
        if (s >= 1 && s <= 11) z = cmac_conj(z, K[j], X[s]);
      }
      Z[c] = z * float2v{zs, zs};
    }
    if (live) {
      float2v* zp = reinterpret_cast<float2v*>(ra + q * TS) + f;  // [column][f], 7 complex per column
#pragma unroll
      for (int c = 0; c < N; ++c) zp[c * NF] = Z[c];
    }
  }

  // ---- stage 3: lane = (plane, column c): rows m = (i + 7) mod 13 and 13 - m share their products ----
  float orow[2][N];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const float2v* zp = reinterpret_cast<const float2v*>(ra + sq[rd] * TS) + ss[rd] * NF;
    float2v Zf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) Zf[f] = zp[f];
    float o0 = Zf[0].x;
#pragma unroll
    for (int f = 1; f < NF; ++f) o0 += Zf[f].x;
    orow[rd][6] = o0;  // m = 0 is row i = 6
#pragma unroll
    for (int m = 1; m <= 6; ++m) {
      float2v ab = float2v{0.f, 0.f};  // (sum Zr cos, sum Zi sin)
#pragma unroll
      for (int f = 1; f < NF; ++f) ab = __builtin_elementwise_fma(Zf[f], tw_inv(tw, f * m), ab);
      orow[rd][(m + 6) % N] = Zf[0].x + ab.x - ab.y;
      orow[rd][(N - m + 6) % N] = Zf[0].x + ab.x + ab.y;
    }
  }
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    if (lane + 64 * rd < np * N) {
      float* os = rb + sq[rd] * XS + ss[rd];
#pragma unroll
      for (int i = 0; i < N; ++i) os[i * N] = orow[rd][i];
    }
  }
  planes_out(rb, P.out[prob] + size_t(p0) * PL, mv, np, lane);
}

// ---------------------------------------------------------------------------------------
// generic runtime-shape kernel: one workgroup per plane
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int circ_row(int r, int HX) {
  int sr = r - HX / 2;
  return sr < 0 ? sr + HX : (sr >= HX ? sr - HX : sr);
}

template <bool USE_LDS>
__global__ __launch_bounds__(HDN_BLOCK) void xcorr_generic_kernel(XcorrPtrs P, int HX, int WX, int HK, int WK,
                                                                   int circ) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int prob = blockIdx.y;
  const int plane = blockIdx.x;
  const int HP = circ ? HX + 2 * (HX / 2) : HX;
  const int WP = circ ? WX + 2 * (WX / 2) : WX;
  const int HO = HP - HK + 1, WO = WP - WK + 1;
  const float* __restrict__ xg = P.x[prob] + size_t(plane) * HX * WX;
  const float* __restrict__ kg = P.k[prob] + size_t(plane) * HK * WK;
  float* __restrict__ og = P.out[prob] + size_t(plane) * HO * WO;

  if constexpr (USE_LDS) {
    float* xs = smem;
    float* ks = smem + HP * WP;
    for (int idx = tid; idx < HP * WP; idx += HDN_BLOCK) {
      const int r = idx / WP, c = idx - r * WP;
      const int sr = circ ? circ_row(r, HX) : r;
      const int sc = circ ? min(max(c - WX / 2, 0), WX - 1) : c;
      xs[idx] = xg[sr * WX + sc];
    }
    for (int idx = tid; idx < HK * WK; idx += HDN_BLOCK) ks[idx] = kg[idx];
    __syncthreads();
    for (int o = tid; o < HO * WO; o += HDN_BLOCK) {
      const int i = o / WO, j = o - i * WO;
      float acc = 0.f;
      for (int u = 0; u < HK; ++u) {
        const float* xr = xs + (i + u) * WP + j;
        const float* kr = ks + u * WK;
        for (int v = 0; v < WK; ++v) acc = __builtin_fmaf(xr[v], kr[v], acc);
      }
      og[o] = acc;
    }
  } else {
    for (int o = tid; o < HO * WO; o += HDN_BLOCK) {
      const int i = o / WO, j = o - i * WO;
      float acc = 0.f;
      for (int u = 0; u < HK; ++u) {
        const int sr = circ ? circ_row(i + u, HX) : i + u;
        for (int v = 0; v < WK; ++v) {
          const int sc = circ ? min(max(j + v - WX / 2, 0), WX - 1) : j + v;
          acc = __builtin_fmaf(xg[sr * WX + sc], kg[u * WK + v], acc);
        }
      }
      og[o] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
static thread_local const char* g_last_variant = "none";

// Variant of the 31x31 (x) 61x61 kernel: set by hdn_xcorr_north_variant(), initially from the environment
// (HDN_NORTH = fft | direct).  Rounds 1-3 carried four more forms (row-first FFT, two-waves-per-SIMD FFT, dense direct sum,
// split-bf16 matrix cores): all parity-green, none faster than these two, retired in round 4 (DESIGN.md section 6 keeps what
// each one taught).
static std::atomic<int> g_north_variant{-1};
static int north_variant() {
  int v = g_north_variant.load(std::memory_order_relaxed);
  if (v >= 0) return v;
  v = HDN_NORTH_FFT_COL;
  const char* e = getenv("HDN_NORTH");
  if (e && e[0] == 'd') v = HDN_NORTH_DIRECT;
  g_north_variant.store(v, std::memory_order_relaxed);
  return v;
}

static int launch_north(const XcorrPtrs& P, int n, int planes, hipStream_t stream) {
  void (*kern)(XcorrPtrs, int) = &xcorr_north_kernel<1>;   // exact-zero taps skipped
  static PerDeviceOnce attr;  // dynamic LDS above 64 KiB needs the opt-in once per kernel and device
  const int dev_ = PerDeviceOnce::device();
  if (!attr.done(dev_)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)north::LDS_BYTES);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr.set(dev_);
  }
  // persistent: 2 workgroups per CU x 256 CUs (fewer if there are fewer planes); n problems share the grid.
  // HDN_NORTH_BLOCKS caps the grid (e.g. 256 = one workgroup per CU, leaving LDS for kernels on other streams).
  static const int cap = [] { const char* e = getenv("HDN_NORTH_BLOCKS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  const int per_problem = max(1, min(cdiv(planes, 4), cap / n));
  hipLaunchKernelGGL(kern, dim3(per_problem, n), dim3(HDN_BLOCK), north::LDS_BYTES, stream, P, planes);
  g_last_variant = "north_61x61_31x31";
  return launch_status();
}

// xcorr_fft.hip
int launch_north_fft(const float* x, const float* k, float* out, int planes, int max_blocks, hipStream_t stream, int pair0);
int launch_north_fft4(const float* x, const float* k, float* out, int planes, int max_blocks, hipStream_t stream);
void disarm_north_launch_events();      // xcorr_fft.hip: hdn_xcorr_north_launch_events is one-shot and must not outlive the call it was armed for



static int launch_prod29(const XcorrPtrs& P, int n, int planes, hipStream_t stream) {
  hipLaunchKernelGGL(xcorr_prod29_kernel, dim3(cdiv(planes, prod29::PPB), n), dim3(HDN_BLOCK), 0, stream, P, planes);
  g_last_variant = "prod_29x29_5x5";
  return launch_status();
}

static int launch_circ13(const XcorrPtrs& P, int n, int planes, hipStream_t stream) {
  // a wave owns a group of 9 planes; the groups of all problems of the launch are one flat grid
  const int gpp = cdiv(planes, circ13f::PPW);
  const long long total = (long long)gpp * n;
  if (total > 0x7fffffffLL) return HDN_E_LIMIT;
  hipLaunchKernelGGL(xcorr_circ13f_kernel, dim3(cdiv((int)total, circ13f::WAVES)), dim3(HDN_BLOCK), 0, stream, P, planes, gpp, (int)total);
  g_last_variant = "circ13";
  return launch_status();
}

static int xcorr_dispatch_(const XcorrPtrs& P, int n, int circular, int B, int C, int Hx, int Wx, int Hk, int Wk, hipStream_t stream);

// Whatever this call launches (or refuses), events armed by hdn_xcorr_north_launch_events do not survive it: they belong to "the next correlation call",
// and a caller is free to destroy them afterwards.
static int xcorr_dispatch(const XcorrPtrs& P, int n, int circular, int B, int C, int Hx, int Wx, int Hk, int Wk, hipStream_t stream) {
  const int rc = xcorr_dispatch_(P, n, circular, B, C, Hx, Wx, Hk, Wk, stream);
  disarm_north_launch_events();
  return rc;
}

static int xcorr_dispatch_(const XcorrPtrs& P, int n, int circular, int B, int C, int Hx, int Wx, int Hk, int Wk,
                           hipStream_t stream) {
  const long long planes_ll = (long long)B * C;
  if (planes_ll > 0x7fffffffLL / 4) return HDN_E_LIMIT;
  const int planes = (int)planes_ll;
  const int HP = circular ? Hx + 2 * (Hx / 2) : Hx, WP = circular ? Wx + 2 * (Wx / 2) : Wx;
  if ((long long)planes * HP * WP > 0x7fffffffLL) return HDN_E_LIMIT;  // 32-bit plane offsets inside a workgroup are
                                                                         // per-block; this bounds the total too
  if (!circular) {
    if (Hx == 29 && Wx == 29 && Hk == 5 && Wk == 5) {
      return launch_prod29(P, n, planes, stream);
    }
    if (Hx == 35 && Wx == 35 && Hk == 5 && Wk == 5) {
      hipLaunchKernelGGL(xcorr_cfg5_kernel, dim3(cdiv(planes, cfg5::PPB), n), dim3(HDN_BLOCK), 0, stream, P, planes);
      g_last_variant = "cfg5_35x35_5x5";
      return launch_status();
    }
    if (Hx == 61 && Wx == 61 && Hk == 31 && Wk == 31) {
      // Default: the column-first FFT kernel (xcorr_fft.hip, ~92 us at B = 64, any pointer alignment); on request
      // (hdn_xcorr_north_variant / HDN_NORTH=direct) the packed-FMA direct sum with zero-tap skipping (~250 us on post-ReLU data).
      const int v = north_variant();
      if (v == HDN_NORTH_FFT_COL) {  // column-first FFT kernel: no alignment requirement
        static const int cap = [] { const char* e = getenv("HDN_NORTH_BLOCKS"); int c = e ? atoi(e) : 0; return c > 0 ? c : 1024; }();
        for (int i = 0; i < n; ++i) {
          const int rc = launch_north_fft4(P.x[i], P.k[i], P.out[i], planes, cap, stream);
          if (rc != HDN_OK) return rc;
        }
        g_last_variant = "north_fftc_61x61_31x31";
        return HDN_OK;
      }
      return launch_north(P, n, planes, stream);
    }
  } else {
    if (Hx == 13 && Wx == 13 && Hk == 13 && Wk == 13) return launch_circ13(P, n, planes, stream);
  }
  const size_t lds = (size_t(HP) * WP + size_t(Hk) * Wk) * sizeof(float);
  dim3 grid(planes, n);
  if (lds <= 60 * 1024) {
    hipLaunchKernelGGL(xcorr_generic_kernel<true>, grid, dim3(HDN_BLOCK), lds, stream, P, Hx, Wx, Hk, Wk, circular);
    g_last_variant = "generic_lds";
  } else {
    hipLaunchKernelGGL(xcorr_generic_kernel<false>, grid, dim3(HDN_BLOCK), 0, stream, P, Hx, Wx, Hk, Wk, circular);
    g_last_variant = "generic_l2";
  }
  return launch_status();
}

static int xcorr_check(int B, int C, int Hx, int Wx, int Hk, int Wk, int circular) {
  if (B <= 0 || C <= 0 || Hx <= 0 || Wx <= 0 || Hk <= 0 || Wk <= 0) return HDN_E_SHAPE;
  const int HP = circular ? Hx + 2 * (Hx / 2) : Hx, WP = circular ? Wx + 2 * (Wx / 2) : Wx;
  if (Hk > HP || Wk > WP) return HDN_E_SHAPE;
  if (Hx > 4096 || Wx > 4096) return HDN_E_LIMIT;
  return HDN_OK;
}

}  // namespace hdn

extern "C" {

const char* hdn_last_xcorr_variant(void) { return hdn::g_last_variant; }

int hdn_xcorr_north_variant(int v) {
  const int prev = hdn::north_variant();
  if (v >= 0) {
    if (v != HDN_NORTH_FFT_COL && v != HDN_NORTH_DIRECT) return HDN_E_LIMIT;
    hdn::g_north_variant.store(v, std::memory_order_relaxed);
  }
  return prev;
}

int hdn_xcorr_depthwise_multi_f32(const float* const* xs, const float* const* ks, float* const* outs, int n,
                                  int circular, int B, int C, int Hx, int Wx, int Hk, int Wk, void* stream) {
  if (!xs || !ks || !outs) return HDN_E_NULL;
  if (n <= 0 || n > hdn::XC_MAX_PROBLEMS) return HDN_E_LIMIT;
  int rc = hdn::xcorr_check(B, C, Hx, Wx, Hk, Wk, circular);
  if (rc) return rc;
  hdn::XcorrPtrs P{};
  for (int i = 0; i < n; ++i) {
    if (!xs[i] || !ks[i] || !outs[i]) return HDN_E_NULL;
    if (outs[i] == xs[i] || outs[i] == ks[i]) return HDN_E_ALIAS;
    P.x[i] = xs[i];
    P.k[i] = ks[i];
    P.out[i] = outs[i];
  }
  return hdn::xcorr_dispatch(P, n, circular, B, C, Hx, Wx, Hk, Wk, static_cast<hipStream_t>(stream));
}

int hdn_xcorr_depthwise_f32(const float* x, const float* k, float* out, int B, int C, int Hx, int Wx, int Hk, int Wk,
                            void* stream) {
  return hdn_xcorr_depthwise_multi_f32(&x, &k, &out, 1, 0, B, C, Hx, Wx, Hk, Wk, stream);
}

int hdn_xcorr_depthwise_circ_f32(const float* x, const float* k, float* out, int B, int C, int Hx, int Wx, int Hk,
                                 int Wk, void* stream) {
  return hdn_xcorr_depthwise_multi_f32(&x, &k, &out, 1, 1, B, C, Hx, Wx, Hk, Wk, stream);
}

}  // extern "C"
