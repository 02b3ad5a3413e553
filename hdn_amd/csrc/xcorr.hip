// Depthwise cross-correlation kernels (plain and log-polar circular) for gfx950.
//
//   out[p,i,j] = sum_{u,v} xp[p,i+u,j+v] * k[p,u,v]        p = b*C + c  ("plane")
//
// Reference semantics: hdn/core/xcorr.py:37-61.  No channel contraction exists in this
// operator, so there is nothing for MFMA to contract; the production shapes are
// HBM-bound (5 FLOP/B) and the 31x31 (x) 61x61 stress shape is fp32-FMA-bound (82 FLOP/B).
//
// Kernel family "f1" (compile-time shapes):
//   * one workgroup = 4 waves = PPB = 4*PPW consecutive planes: their x planes are ONE
//     contiguous HBM range, fetched with 16-byte coalesced loads into LDS (linear image
//     when SX == WX, else re-strided so 16-byte LDS reads stay aligned and conflict-free);
//   * one wave owns one plane at a time; its HKxWK taps are wave-uniform and are read
//     through the scalar cache into SGPRs (no LDS/VGPR cost, FMA takes the SGPR operand);
//   * a lane owns a 1 x TW strip of outputs: per tap row it reads TW+WK-1 floats from LDS
//     and issues TW*WK FMAs, accumulating in a fixed (u,v) order (deterministic);
//   * results are staged in LDS and leave as one contiguous 16-byte coalesced store.
// Circular variant: the padded plane (rows wrap, columns clamp) is built in LDS only.
//
// Kernel "generic" (runtime shapes): one workgroup per plane, LDS-staged when the plane
// fits, straight from L2 otherwise.  Correct for any Hk<=Hx, Wk<=Wx; not tuned.
#include "hdn_common.h"

namespace hdn {

constexpr int XC_MAX_PROBLEMS = 8;

struct XcorrPtrs {
  const float* x[XC_MAX_PROBLEMS];
  const float* k[XC_MAX_PROBLEMS];
  float* out[XC_MAX_PROBLEMS];
};

template <int HX_, int WX_, int HK_, int WK_, int TW_, int SX_, int PPW_, bool CIRC_, bool REUSE_>
struct F1Cfg {
  static constexpr int HX = HX_, WX = WX_, HK = HK_, WK = WK_, TW = TW_, SX = SX_, PPW = PPW_;
  static constexpr bool CIRC = CIRC_, REUSE = REUSE_;
  static constexpr int HP = CIRC ? HX + 2 * (HX / 2) : HX;  // plane as correlated (padded if circular)
  static constexpr int WP = CIRC ? WX + 2 * (WX / 2) : WX;
  static constexpr int HO = HP - HK + 1, WO = WP - WK + 1;
  static constexpr int NSEG = cdiv(WO, TW);
  static constexpr int UNITS = HO * NSEG;      // strips per plane
  static constexpr int XW = TW + WK - 1;       // LDS floats a strip reads per tap row
  static constexpr int PPB = 4 * PPW;          // planes per workgroup
  static constexpr int XPLANE = HP * SX;       // LDS floats per staged plane
  static constexpr int OPLANE = HO * WO;
  static constexpr int SLACK = 64;             // the last strip of the last row may over-read (never stored)
  static constexpr int XFLOATS = round_up(PPB * XPLANE + SLACK, 4);
  static constexpr int LDS_FLOATS = XFLOATS + (REUSE ? 0 : round_up(PPB * OPLANE, 4));
  static constexpr size_t LDS_BYTES = size_t(LDS_FLOATS) * sizeof(float);
  static_assert(SX >= WP, "row stride shorter than the plane");
  static_assert(!REUSE || UNITS <= HDN_WAVE, "output staging may reuse the x region only with one strip round");
  static_assert(!REUSE || OPLANE <= XPLANE, "output does not fit the reused region");
  static_assert((NSEG - 1) * TW + XW <= SX + SLACK, "strip over-read exceeds the slack");
};

template <class Cfg>
__global__ __launch_bounds__(HDN_BLOCK) void xcorr_f1_kernel(XcorrPtrs P, int planes) {
  constexpr int HX = Cfg::HX, WX = Cfg::WX, HK = Cfg::HK, WK = Cfg::WK, TW = Cfg::TW, SX = Cfg::SX;
  constexpr int HP = Cfg::HP, WP = Cfg::WP, WO = Cfg::WO, NSEG = Cfg::NSEG, UNITS = Cfg::UNITS;
  constexpr int XW = Cfg::XW, PPW = Cfg::PPW, PPB = Cfg::PPB, XPLANE = Cfg::XPLANE, OPLANE = Cfg::OPLANE;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sx = smem;
  float* so = Cfg::REUSE ? smem : smem + Cfg::XFLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & (HDN_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prob = blockIdx.y;
  const float* __restrict__ x = P.x[prob];
  const float* __restrict__ k = P.k[prob];
  float* __restrict__ out = P.out[prob];

  const int plane0 = blockIdx.x * PPB;
  const int np = min(PPB, planes - plane0);

  // ---- stage the x planes of this workgroup in LDS ------------------------------------
  const float* xg = x + size_t(plane0) * (HX * WX);
  if constexpr (!Cfg::CIRC && SX == WX) {
    if (np == PPB && aligned16(xg)) copy_g2l_full<PPB * HX * WX>(xg, sx, tid);  // all loads in flight at once
    else copy_g2l(xg, sx, np * HX * WX, tid);
  } else if constexpr (!Cfg::CIRC) {
    for (int idx = tid; idx < np * HX * WX; idx += HDN_BLOCK) {
      const int p = idx / (HX * WX), rem = idx - p * (HX * WX);
      const int r = rem / WX, c = rem - r * WX;
      sx[p * XPLANE + r * SX + c] = xg[idx];
    }
  } else {
    // rows wrap by HX/2 (angle axis), columns clamp by WX/2 (log-radius axis): xcorr.py:52-53.
    // Re-reads hit L1/L2; HBM sees each x element once.
    for (int idx = tid; idx < np * HP * WP; idx += HDN_BLOCK) {
      const int p = idx / (HP * WP), rem = idx - p * (HP * WP);
      const int r = rem / WP, c = rem - r * WP;
      int sr = r - HX / 2;
      sr = sr < 0 ? sr + HX : (sr >= HX ? sr - HX : sr);
      const int sc = min(max(c - WX / 2, 0), WX - 1);
      sx[p * XPLANE + r * SX + c] = xg[p * (HX * WX) + sr * WX + sc];
    }
  }
  __syncthreads();

  // ---- correlate: one wave per plane, one lane per 1 x TW output strip -----------------
#pragma unroll 1
  for (int pw = 0; pw < PPW; ++pw) {
    const int slot = wave * PPW + pw;  // wave-uniform
    if (slot < np) {
      const float* __restrict__ kp = k + size_t(plane0 + slot) * (HK * WK);  // wave-uniform -> scalar loads
      const float* xs = sx + slot * XPLANE;
      float* os = so + slot * (Cfg::REUSE ? XPLANE : OPLANE);
#pragma unroll 1
      for (int unit0 = 0; unit0 < UNITS; unit0 += HDN_WAVE) {
        const int unit = unit0 + lane;
        const bool live = unit < UNITS;
        const int uu = live ? unit : UNITS - 1;
        const int i = uu / NSEG, s = uu - i * NSEG;
        const float* xr = xs + i * SX + s * TW;
        float acc[TW];
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[j] = 0.f;

        if constexpr (HK * WK <= 64) {
#pragma unroll
          for (int u = 0; u < HK; ++u) {
            float xv[XW];
#pragma unroll
            for (int c = 0; c < XW; ++c) xv[c] = xr[u * SX + c];
#pragma unroll
            for (int v = 0; v < WK; ++v) {
              const float kv = kp[u * WK + v];
#pragma unroll
              for (int j = 0; j < TW; ++j) acc[j] = __builtin_fmaf(xv[j + v], kv, acc[j]);
            }
          }
        } else {
#pragma unroll 1
          for (int u = 0; u < HK; ++u) {
            float xv[XW];
#pragma unroll
            for (int c = 0; c < XW; ++c) xv[c] = xr[u * SX + c];
#pragma unroll
            for (int v = 0; v < WK; ++v) {
              const float kv = kp[u * WK + v];
#pragma unroll
              for (int j = 0; j < TW; ++j) acc[j] = __builtin_fmaf(xv[j + v], kv, acc[j]);
            }
          }
        }
        if constexpr (Cfg::REUSE) __builtin_amdgcn_wave_barrier();  // every lane's x reads precede the overwrite
        if (live) {
#pragma unroll
          for (int j = 0; j < TW; ++j)
            if (s * TW + j < WO) os[i * WO + s * TW + j] = acc[j];
        }
      }
    }
  }
  __syncthreads();

  // ---- contiguous store of the workgroup's output planes -------------------------------
  float* og = out + size_t(plane0) * OPLANE;
  if constexpr (!Cfg::REUSE) {
    if (np == PPB && aligned16(og)) copy_l2g_full<PPB * OPLANE>(so, og, tid);
    else copy_l2g(so, og, np * OPLANE, tid);
  } else {
    for (int idx = tid; idx < np * OPLANE; idx += HDN_BLOCK) {
      const int p = idx / OPLANE, rem = idx - p * OPLANE;
      og[idx] = so[p * XPLANE + rem];
    }
  }
}

// ---------------------------------------------------------------------------------------
// 31x31 (x) 61x61 -> 31x31 (BASELINE.json north-star shape).  fp32-FMA-bound: 1.85 MFLOP per 23 KB plane.
//
// gfx950 only reaches its fp32 vector peak through v_pk_fma_f32 (measured: 151 TF vs 75 TF for v_fma_f32,
// profiles/round1_ubench_fma.txt), so the kernel is built around packed FMAs with no register shuffles:
//   * a lane owns 16 outputs of one row as 8 pairs (out[j], out[j+8]); the matching operand pair
//     (x[c], x[c+8]) is ONE ds_read2_b32, so every tap is `acc[j] += P[j+v] * k[v]` on aligned pairs,
//     whatever the parity of v (adjacent-column pairs would need a shifted copy for odd v);
//   * the plane sits in LDS as a linear image (row stride 61, odd => conflict-free b32 reads) filled by
//     16-byte loads that are all in flight at once;
//   * the 31 taps of a kernel row are wave-uniform: scalar loads into SGPRs, broadcast by op_sel;
//   * tap row u+1 (38 LDS pairs + 31 SGPRs) is fetched while row u's 248 packed FMAs issue.
// ---------------------------------------------------------------------------------------
namespace north {
constexpr int HX = 61, WX = 61, HK = 31, WK = 31, HO = 31, WO = 31;
constexpr int PPB = 4;                  // planes per workgroup = one per wave
constexpr int XPLANE = HX * WX;         // 3721, linear image
constexpr int OPLANE = HO * WO;         // 961
constexpr int NPAIR = 38;               // (x[c], x[c+8]) for c = 0..37 covers 16 outputs x 31 taps
constexpr int XFLOATS = round_up(PPB * XPLANE + 64, 4);  // + slack: the last row's pairs over-read <= 8 floats
constexpr int LDS_FLOATS = XFLOATS + PPB * OPLANE;
constexpr size_t LDS_BYTES = size_t(LDS_FLOATS) * sizeof(float);

struct Row {
  float2v P[NPAIR];
  float k[WK];
};

template <int C>
__device__ __forceinline__ void issue_pairs(Row& R, uint32_t a) {
  if constexpr (C < NPAIR) {
    R.P[C] = lds_read_pair<C, C + 8>(a);
    issue_pairs<C + 1>(R, a);
  }
}

// Issue the 38 LDS pair reads and the 31 scalar tap loads of one kernel row.  Nothing is waited for here.
__device__ __forceinline__ void load_row(Row& R, uint32_t xaddr, const float* __restrict__ kr) {
  issue_pairs<0>(R, xaddr);
#pragma unroll
  for (int v = 0; v < WK; ++v) R.k[v] = kr[v];
}

// All outstanding LDS reads have landed; pin the pairs so no FMA that reads them floats above the wait.
__device__ __forceinline__ void land_row(Row& R) {
  lds_wait_all();
#pragma unroll
  for (int c = 0; c < NPAIR; ++c) pin(R.P[c]);
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void fma_row(float2v (&acc)[8], const Row& R) {
#pragma unroll
  for (int v = 0; v < WK; ++v) {
    const float2v kk = {R.k[v], R.k[v]};
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_elementwise_fma(R.P[j + v], kk, acc[j]);
  }
}
}  // namespace north

__global__ __launch_bounds__(HDN_BLOCK, 2) void xcorr_north_kernel(XcorrPtrs P, int planes) {
  using namespace north;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sx = smem;
  float* so = smem + XFLOATS;

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
  if (np == PPB && aligned16(xg)) copy_g2l_full<PPB * XPLANE>(xg, sx, tid);
  else copy_g2l(xg, sx, np * XPLANE, tid);
  __syncthreads();

  if (wave < np) {  // wave-uniform
    const float* __restrict__ kp = k + size_t(plane0 + wave) * (HK * WK);
    const int i = min(lane & 31, HO - 1);  // output row (lanes 31 and 63 shadow row 30 and do not store)
    const int s = lane >> 5;               // output columns [16 s, 16 s + 16)
    const uint32_t xa = lds_addr(sx + wave * XPLANE + i * WX + s * 16);
    float2v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = float2v{0.f, 0.f};
    Row A, B;
    load_row(A, xa, kp);
    land_row(A);
#pragma unroll 1
    for (int u = 0; u < HK - 1; u += 2) {
      load_row(B, xa + (u + 1) * (WX * 4), kp + (u + 1) * WK);
      __builtin_amdgcn_sched_barrier(0);
      fma_row(acc, A);
      land_row(B);
      load_row(A, xa + (u + 2) * (WX * 4), kp + (u + 2) * WK);
      __builtin_amdgcn_sched_barrier(0);
      fma_row(acc, B);
      land_row(A);
    }
    fma_row(acc, A);  // u = 30
    if ((lane & 31) < HO) {
      float* os = so + wave * OPLANE + i * WO + s * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        os[j] = acc[j].x;
        if (s * 16 + j + 8 < WO) os[j + 8] = acc[j].y;
      }
    }
  }
  __syncthreads();
  float* og = out + size_t(plane0) * OPLANE;
  if (np == PPB && aligned16(og)) copy_l2g_full<PPB * OPLANE>(so, og, tid);
  else copy_l2g(so, og, np * OPLANE, tid);
}

// ---------------------------------------------------------------------------------------
// circular 13x13 (x) 13x13 -> 13x13 (log-polar head, ban_lp.py:38).  ~ridge: 28 FLOP/B.
//
// The 25x25 padded plane is never built: x and k stay in LDS as raw 13x13 linear images (both are contiguous
// 16-plane chunks in HBM -> two all-in-flight 16-byte copies), and the wrap / clamp is folded into addresses:
//   padded row  (i+u)  -> source row (i+u+7) mod 13      (rows = angle, wraps)
//   padded col  c      -> source col clamp(c-6, 0, 12)   (cols = log-radius, replicates) : a compile-time index
// A lane owns one output row of one plane (13 lanes per plane, 4 planes per wave) as 7 pairs (out[j], out[j+7]);
// the operand pair (xp[c], xp[c+7]) is one ds_read2_b32 with two static column offsets, so all taps are packed
// FMAs on aligned pairs.  k rows are read from LDS (same address across a plane's 13 lanes: broadcast).
// ---------------------------------------------------------------------------------------
namespace circ13 {
constexpr int N = 13, PL = N * N;       // 169
constexpr int PPW = 4, PPB = 16;        // planes per wave / per workgroup
constexpr int NPAIR = 19;               // (xp[c], xp[c+7]), c = 0..18
constexpr int LDS_FLOATS = 3 * PPB * PL + 16;
__host__ __device__ constexpr int clampc(int c) { return c < 6 ? 0 : (c > 18 ? 12 : c - 6); }
template <int C>
__device__ __forceinline__ void issue_pairs(float2v (&Pp)[NPAIR], uint32_t a) {
  if constexpr (C < NPAIR) {
    Pp[C] = lds_read_pair<clampc(C), clampc(C + 7)>(a);
    issue_pairs<C + 1>(Pp, a);
  }
}
}  // namespace circ13

__global__ __launch_bounds__(HDN_BLOCK) void xcorr_circ13_kernel(XcorrPtrs P, int planes) {
  using namespace circ13;
  __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
  float* sx = smem;
  float* sk = smem + PPB * PL;
  float* so = smem + 2 * PPB * PL;

  const int tid = threadIdx.x;
  const int lane = tid & (HDN_WAVE - 1);
  const int wave = tid >> 6;
  const int prob = blockIdx.y;
  const int plane0 = blockIdx.x * PPB;
  const int np = min(PPB, planes - plane0);
  const float* xg = P.x[prob] + size_t(plane0) * PL;
  const float* kg = P.k[prob] + size_t(plane0) * PL;
  float* og = P.out[prob] + size_t(plane0) * PL;

  if (np == PPB && aligned16(xg) && aligned16(kg)) {
    copy_g2l_full<PPB * PL>(xg, sx, tid);
    copy_g2l_full<PPB * PL>(kg, sk, tid);
  } else {
    copy_g2l(xg, sx, np * PL, tid);
    copy_g2l(kg, sk, np * PL, tid);
  }
  __syncthreads();

  const int q = min(lane / N, PPW - 1);       // lanes 52..63 shadow plane 3 and do not store
  const int i = lane - (lane / N) * N;        // output row
  const int slot = min(wave * PPW + q, np - 1);
  const bool live = lane < PPW * N && wave * PPW + q < np;
  const float* xs = sx + slot * PL;
  const float* ks = sk + slot * PL;
  float2v acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[j] = float2v{0.f, 0.f};
  int r = i + 7;  // source row of padded row i + u, u = 0
  r = r >= N ? r - N : r;
#pragma unroll 1
  for (int u = 0; u < N; ++u) {
    const float* xr = xs + r * N;
    const float* kr = ks + u * N;
    float2v Pp[NPAIR];
    circ13::issue_pairs<0>(Pp, lds_addr(xr));
    float kv_[N];
#pragma unroll
    for (int v = 0; v < N; ++v) kv_[v] = kr[v];
    lds_wait_all();
#pragma unroll
    for (int c = 0; c < NPAIR; ++c) pin(Pp[c]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < N; ++v) {
      const float2v kk = {kv_[v], kv_[v]};
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = __builtin_elementwise_fma(Pp[j + v], kk, acc[j]);
    }
    r = (r + 1 == N) ? 0 : r + 1;
  }
  if (live) {
    float* os = so + slot * PL + i * N;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      os[j] = acc[j].x;
      if (j + 7 < N) os[j + 7] = acc[j].y;
    }
  }
  __syncthreads();
  if (np == PPB && aligned16(og)) copy_l2g_full<PPB * PL>(so, og, tid);
  else copy_l2g(so, og, np * PL, tid);
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

template <class Cfg>
static int launch_f1(const XcorrPtrs& P, int n, int planes, hipStream_t stream, const char* name) {
  static bool attr_done = false;  // dynamic LDS above 64 KiB needs the opt-in once per kernel
  if (!attr_done) {
    if (Cfg::LDS_BYTES > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xcorr_f1_kernel<Cfg>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
      if (e != hipSuccess) return -(1000 + (int)e);
    }
    attr_done = true;
  }
  dim3 grid(cdiv(planes, Cfg::PPB), n);
  hipLaunchKernelGGL(xcorr_f1_kernel<Cfg>, grid, dim3(HDN_BLOCK), Cfg::LDS_BYTES, stream, P, planes);
  g_last_variant = name;
  return launch_status();
}

static int launch_north(const XcorrPtrs& P, int n, int planes, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xcorr_north_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)north::LDS_BYTES);
    if (e != hipSuccess) return -(1000 + (int)e);
    attr_done = true;
  }
  hipLaunchKernelGGL(xcorr_north_kernel, dim3(cdiv(planes, north::PPB), n), dim3(HDN_BLOCK), north::LDS_BYTES, stream, P,
                     planes);
  g_last_variant = "north_61x61_31x31";
  return launch_status();
}

static int launch_circ13(const XcorrPtrs& P, int n, int planes, hipStream_t stream) {
  hipLaunchKernelGGL(xcorr_circ13_kernel, dim3(cdiv(planes, circ13::PPB), n), dim3(HDN_BLOCK), 0, stream, P, planes);
  g_last_variant = "circ13";
  return launch_status();
}

//                 HX  WX  HK  WK  TW  SX  PPW  CIRC   REUSE
using F1_29_5 = F1Cfg<29, 29, 5, 5, 5, 29, 2, false, false>;      // production: 3 levels x {cls,loc}, ban.py:76
using F1_35_5 = F1Cfg<35, 35, 5, 5, 8, 35, 2, false, false>;      // INSTANCE_SIZE 303 (BASELINE config 5)

static int xcorr_dispatch(const XcorrPtrs& P, int n, int circular, int B, int C, int Hx, int Wx, int Hk, int Wk,
                          hipStream_t stream) {
  const long long planes_ll = (long long)B * C;
  if (planes_ll > 0x7fffffffLL / 4) return HDN_E_LIMIT;
  const int planes = (int)planes_ll;
  const int HP = circular ? Hx + 2 * (Hx / 2) : Hx, WP = circular ? Wx + 2 * (Wx / 2) : Wx;
  if ((long long)planes * HP * WP > 0x7fffffffLL) return HDN_E_LIMIT;  // 32-bit plane offsets inside a workgroup are
                                                                         // per-block; this bounds the total too
  if (!circular) {
    if (Hx == 29 && Wx == 29 && Hk == 5 && Wk == 5) return launch_f1<F1_29_5>(P, n, planes, stream, "f1_29x29_5x5");
    if (Hx == 35 && Wx == 35 && Hk == 5 && Wk == 5) return launch_f1<F1_35_5>(P, n, planes, stream, "f1_35x35_5x5");
    if (Hx == 61 && Wx == 61 && Hk == 31 && Wk == 31) return launch_north(P, n, planes, stream);
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
