// pack.hip — the weight streams of the matrix-core kernels, built by the LIBRARY (host code, no kernel in this file).
//
// Until ABI 9 the layouts below were part of the ABI: a host had to split fp32 weights into two fp16 pieces and lay them out in the fragment
// order of the kernel it was going to call (include/hdn_hip.h described every index), so every retune of a tile shape broke C hosts.  Since
// ABI 10 a host hands over plain fp32 weights in the reference's own order ([CO][CI][kh][kw], what a state_dict holds) and gets back an opaque
// stream for the matching entry point; how the stream is ordered is this file's business and the kernels'.
//
//   hdn_pack_conv3x3_f32        -> wpacked of hdn_conv3x3_bias_relu_f32 / hdn_conv3x3_chain_f32        (Conv2d(C, C, 3, 1, 1) of a BasicBlock)
//   hdn_pack_conv3x3s2_ds_f32   -> wpacked of hdn_conv3x3s2_ds_f32                                     (stride-2 conv + the 1x1 downsample branch)
//   hdn_pack_conv3x3_v2_f32     -> wpacked of hdn_conv3x3_v2_f32
//   hdn_pack_conv3x3s2_v2_f32   -> wpacked of hdn_conv3x3s2_v2_f32
//   hdn_pack_stem_mfma_f32      -> wfrag of hdn_trunk_stem_mfma_f32                                    (Conv2d(2, 64, 7, 2, 3))
//   hdn_pack_head_conv3x3_f32   -> w_packed of hdn_head_conv3x3_f32                                    (n x Conv2d(256, CO, 3))
//   hdn_pack_head_tail_f32      -> w1_packed of hdn_head_tail_f32                                      (G x [H, H] 1x1 convolutions)
//
// All pointers are HOST pointers.  Every stream is 2 pieces x 2 bytes per (padded) weight: v = p0 + 2^-11 p1, p0 = fp16(v),
// p1 = fp16((v - p0) 2^11), round-to-nearest-even each (csrc/mfma_split.h; weights are NOT pre-scaled, only activations are).
// hdn_pack_*_bytes() gives the size; a weight with |w| >= 65,504 or NaN -> HDN_E_LIMIT (the packer is where weights are range-checked).
// BatchNorm folding stays the caller's (it is arithmetic on the model, not a layout).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/hdn_hip.h"

namespace {

inline bool split(float v, uint16_t& p0, uint16_t& p1) {
  if (!(std::fabs(v) < 65504.0f)) return false;          // (also false for NaN)
  const _Float16 h = static_cast<_Float16>(v);
  const float r = (v - static_cast<float>(h)) * 2048.0f;
  const _Float16 l = static_cast<_Float16>(r);
  std::memcpy(&p0, &h, 2);
  std::memcpy(&p1, &l, 2);
  return true;
}

// out[((...) * 2 + piece) ...]: every layout below is "some index order with the piece index somewhere in the middle"; the callers compute
// the two destination offsets themselves and hand the value over.
struct Sink {
  uint16_t* out;
  bool ok = true;
  inline void put(long long off0, long long off1, float v) {
    uint16_t a, b;
    if (!split(v, a, b)) { ok = false; a = b = 0; }
    out[off0] = a;
    out[off1] = b;
  }
};

// [CO / BN][CI / (16 KS)][3][T][KS][2 pieces][2 k halves][BN][8]: the stream of conv3x3_kernel (T = 3) and of its stride-2 form (T = 4)
int pack_taps(const float* w3, const float* ds, int CO, int CI, int T, int BN, int KS, uint16_t* out) {
  Sink s{out};
  const int chunks = CI / (16 * KS);
  for (int co = 0; co < CO; ++co)
    for (int ci = 0; ci < CI; ++ci)
      for (int ky = 0; ky < 3; ++ky)
        for (int t = 0; t < T; ++t) {
          float v;
          if (t < 3) v = w3[((static_cast<long long>(co) * CI + ci) * 3 + ky) * 3 + t];
          else v = (ky == 1 && ds) ? ds[static_cast<long long>(co) * CI + ci] : 0.0f;      // the 1x1 branch rides as a 4th tap of the middle kernel row
          const int nb = co / BN, n = co % BN, chunk = ci / (16 * KS), ks = (ci / 16) % KS, g = (ci / 8) & 1, e = ci & 7;
          long long o = ((((static_cast<long long>(nb) * chunks + chunk) * 3 + ky) * T + t) * KS + ks) * 2;   // ... [piece]
          const long long tail = 2LL * BN * 8;                                                               // [g][n][8] per piece
          s.put((o + 0) * tail + (static_cast<long long>(g) * BN + n) * 8 + e, (o + 1) * tail + (static_cast<long long>(g) * BN + n) * 8 + e, v);
        }
  return s.ok ? HDN_OK : HDN_E_LIMIT;
}

const int kSide[4][2] = {{64, 32}, {128, 16}, {256, 8}, {512, 4}};      // channels -> output side of the trunk's stages at 127-px crops
int side_of(int C) {
  for (auto& p : kSide)
    if (p[0] == C) return p[1];
  return 0;
}

}  // namespace

extern "C" long long hdn_pack_conv3x3_bytes(int C) { return side_of(C) ? 2LL * 2 * C * C * 9 : HDN_E_SHAPE; }

extern "C" int hdn_pack_conv3x3_f32(const float* w, int C, void* out, long long out_bytes) {
  if (!w || !out) return HDN_E_NULL;
  int BN = 0, KS = 0;
  if (!side_of(C) || hdn_conv3x3_pack_info(side_of(C), C, 1, &BN, &KS) != HDN_OK) return HDN_E_SHAPE;
  if (out_bytes != hdn_pack_conv3x3_bytes(C)) return HDN_E_SHAPE;
  return pack_taps(w, nullptr, C, C, 3, BN, KS, static_cast<uint16_t*>(out));
}

extern "C" long long hdn_pack_conv3x3s2_ds_bytes(int CI) { return side_of(2 * CI) ? 2LL * 2 * (2 * CI) * CI * 12 : HDN_E_SHAPE; }

extern "C" int hdn_pack_conv3x3s2_ds_f32(const float* w, const float* w_ds, int CI, void* out, long long out_bytes) {
  if (!w || !w_ds || !out) return HDN_E_NULL;
  int BN = 0, KS = 0;
  const int CO = 2 * CI;
  if (!side_of(CO) || hdn_conv3x3_pack_info(side_of(CO), CI, 2, &BN, &KS) != HDN_OK) return HDN_E_SHAPE;
  if (out_bytes != hdn_pack_conv3x3s2_ds_bytes(CI)) return HDN_E_SHAPE;
  return pack_taps(w, w_ds, CO, CI, 4, BN, KS, static_cast<uint16_t*>(out));
}

extern "C" long long hdn_pack_conv3x3_v2_bytes(int C) { return side_of(C) ? 2LL * 2 * C * C * 9 : HDN_E_SHAPE; }

// [C / (32 NT)][chunk][k slice][9 taps][k step of the slice][n tile][piece][k half][32 n][8]
extern "C" int hdn_pack_conv3x3_v2_f32(const float* w, int C, void* out, long long out_bytes) {
  if (!w || !out) return HDN_E_NULL;
  int WK = 0, KS = 0, NT = 0;
  if (!side_of(C) || hdn_conv3x3_v2_pack_info(side_of(C), C, &WK, &KS, &NT) != HDN_OK) return HDN_E_SHAPE;
  if (out_bytes != hdn_pack_conv3x3_v2_bytes(C)) return HDN_E_SHAPE;
  Sink s{static_cast<uint16_t*>(out)};
  const int chunks = C / (16 * KS), J = KS / WK;
  for (int co = 0; co < C; ++co)
    for (int ci = 0; ci < C; ++ci)
      for (int tap = 0; tap < 9; ++tap) {
        const int nb = co / (32 * NT), nt = (co / 32) % NT, n = co & 31;
        const int ch = ci / (16 * KS), r = (ci / 16) % KS, j = r / WK, wk = r % WK, g = (ci / 8) & 1, e = ci & 7;   // ci = ch 16 KS + (j WK + wk) 16 + g 8 + e
        const long long o = (((((static_cast<long long>(nb) * chunks + ch) * WK + wk) * 9 + tap) * J + j) * NT + nt) * 2;
        const long long tail = 2LL * 32 * 8;
        const long long in = (static_cast<long long>(g) * 32 + n) * 8 + e;
        s.put((o + 0) * tail + in, (o + 1) * tail + in, w[(static_cast<long long>(co) * C + ci) * 9 + tap]);
      }
  return s.ok ? HDN_OK : HDN_E_LIMIT;
}

extern "C" long long hdn_pack_conv3x3s2_v2_bytes(int CI) { return (CI > 0 && CI % 32 == 0) ? 2LL * 2 * (2 * CI) * CI * 10 : HDN_E_SHAPE; }

// [2C / 64][C / 32 chunks][2 k steps][10 steps: nine taps + the downsample branch][2 n tiles][piece][k half][32 n][8]
extern "C" int hdn_pack_conv3x3s2_v2_f32(const float* w, const float* w_ds, int CI, void* out, long long out_bytes) {
  if (!w || !w_ds || !out) return HDN_E_NULL;
  if (CI <= 0 || CI % 32 || out_bytes != hdn_pack_conv3x3s2_v2_bytes(CI)) return HDN_E_SHAPE;
  const int CO = 2 * CI, chunks = CI / 32;
  Sink s{static_cast<uint16_t*>(out)};
  for (int co = 0; co < CO; ++co)
    for (int ci = 0; ci < CI; ++ci)
      for (int step = 0; step < 10; ++step) {
        const float v = step < 9 ? w[(static_cast<long long>(co) * CI + ci) * 9 + step] : w_ds[static_cast<long long>(co) * CI + ci];
        const int nb = co / 64, nt = (co / 32) & 1, n = co & 31, chunk = ci / 32, wk = (ci / 16) & 1, g = (ci / 8) & 1, e = ci & 7;
        const long long o = ((((static_cast<long long>(nb) * chunks + chunk) * 2 + wk) * 10 + step) * 2 + nt) * 2;
        const long long tail = 2LL * 32 * 8, in = (static_cast<long long>(g) * 32 + n) * 8 + e;
        s.put((o + 0) * tail + in, (o + 1) * tail + in, v);
      }
  return s.ok ? HDN_OK : HDN_E_LIMIT;
}

extern "C" long long hdn_pack_stem_mfma_bytes(void) { return 2LL * 2 * 64 * 14 * 8; }

// w [64][2][7][7] -> [7 k steps][2 n tiles][2 pieces][k half g][32 n][8]: element j = w[32 tile + n][ci][ky][kx = j], ci * 7 + ky = 2 step + g, zero at j = 7
extern "C" int hdn_pack_stem_mfma_f32(const float* w, void* out, long long out_bytes) {
  if (!w || !out) return HDN_E_NULL;
  if (out_bytes != hdn_pack_stem_mfma_bytes()) return HDN_E_SHAPE;
  Sink s{static_cast<uint16_t*>(out)};
  for (int co = 0; co < 64; ++co)
    for (int r = 0; r < 14; ++r)
      for (int j = 0; j < 8; ++j) {
        const float v = j < 7 ? w[(static_cast<long long>(co) * 14 + r) * 7 + j] : 0.0f;
        const int tile = co / 32, n = co & 31, step = r / 2, g = r & 1;
        const long long o = (static_cast<long long>(step) * 2 + tile) * 2;
        const long long tail = 2LL * 32 * 8, in = (static_cast<long long>(g) * 32 + n) * 8 + j;
        s.put((o + 0) * tail + in, (o + 1) * tail + in, v);
      }
  return s.ok ? HDN_OK : HDN_E_LIMIT;
}

extern "C" long long hdn_pack_head_conv3x3_bytes(int n, int CO) { return (n > 0 && CO > 0 && CO % 32 == 0) ? 2LL * 2 * n * CO * 256 * 9 : HDN_E_SHAPE; }

// n x [CO][256][3][3] -> [n][CO / 32][4 chunks][4 k slices][9 taps][piece][k half][32 output channels][8]
extern "C" int hdn_pack_head_conv3x3_f32(const float* const* ws, int n, int CO, void* out, long long out_bytes) {
  if (!ws || !out) return HDN_E_NULL;
  if (n <= 0 || CO <= 0 || CO % 32 || out_bytes != hdn_pack_head_conv3x3_bytes(n, CO)) return HDN_E_SHAPE;
  const int CI = 256;
  Sink s{static_cast<uint16_t*>(out)};
  for (int l = 0; l < n; ++l) {
    if (!ws[l]) return HDN_E_NULL;
    for (int co = 0; co < CO; ++co)
      for (int ci = 0; ci < CI; ++ci)
        for (int tap = 0; tap < 9; ++tap) {
          const int cb = co / 32, m = co & 31, chunk = ci / 64, sl = (ci / 16) & 3, g = (ci / 8) & 1, e = ci & 7;
          const long long o = ((((static_cast<long long>(l) * (CO / 32) + cb) * 4 + chunk) * 4 + sl) * 9 + tap) * 2;
          const long long tail = 2LL * 32 * 8, in = (static_cast<long long>(g) * 32 + m) * 8 + e;
          s.put((o + 0) * tail + in, (o + 1) * tail + in, ws[l][(static_cast<long long>(co) * CI + ci) * 9 + tap]);
        }
  }
  return s.ok ? HDN_OK : HDN_E_LIMIT;
}

extern "C" long long hdn_pack_head_tail_bytes(int G, int H) { return (G > 0 && H > 0 && H % 32 == 0) ? 2LL * 2 * G * H * H : HDN_E_SHAPE; }

// w1 [G][H][H] (row = output channel) -> [G][H / 32 m tiles][H / 16 k steps][piece][k half][32 rows][8]
extern "C" int hdn_pack_head_tail_f32(const float* w1, int G, int H, void* out, long long out_bytes) {
  if (!w1 || !out) return HDN_E_NULL;
  if (G <= 0 || H <= 0 || H % 32 || out_bytes != hdn_pack_head_tail_bytes(G, H)) return HDN_E_SHAPE;
  Sink s{static_cast<uint16_t*>(out)};
  for (int g = 0; g < G; ++g)
    for (int row = 0; row < H; ++row)
      for (int k = 0; k < H; ++k) {
        const int mt = row / 32, r = row & 31, kstep = k / 16, kh = (k / 8) & 1, j = k & 7;
        const long long o = ((static_cast<long long>(g) * (H / 32) + mt) * (H / 16) + kstep) * 2;
        const long long tail = 2LL * 32 * 8, in = (static_cast<long long>(kh) * 32 + r) * 8 + j;
        s.put((o + 0) * tail + in, (o + 1) * tail + in, w1[(static_cast<long long>(g) * H + row) * H + k]);
      }
  return s.ok ? HDN_OK : HDN_E_LIMIT;
}
