// Decode of the similarity branch's head maps on the device (BASELINE configs[3]).
//
//   hdn_similarity_translation_f32   hdnTrackerHomo.track_new, hdn/tracker/hdn_tracker_proj_e2e.py:169-186:
//                                    _convert_score (hdn/tracker/hdn_tracker.py:84-91), _convert_c (hdn/tracker/base_tracker.py:54-59),
//                                    Hanning-window blend (cfg.TRACK.WINDOW_INFLUENCE), argmax, the 0.05 gate, centre shift
//   hdn_similarity_logpolar_f32      :197-214: _convert_score of the log-polar head, argmax, _convert_logpolar_simi
//                                    (hdn_tracker.py:51-67), the 0.25 gate, scale_delta / rot_delta, H_sim =
//                                    rot_scale_around_center_shift_tran (hdn/utils/transform.py:250-298), plus what the rest of the
//                                    frame needs from them: the rotate-back matrix of img_rot_around_center (:223, transform.py:69-100)
//                                    and the crop parameters of get_subwindow_for_homo (:224-227)
//
// The reference does this in numpy after four .cpu().numpy() reads per frame; here the maps never leave the device and the
// results land in a small float64 state record that the crop / warp kernels of frame.hip read directly, so the whole frame
// stays capturable in one hipGraph.  One wave per pair; dtypes follow the reference statement by statement (float32 scores and
// regression values, float64 window blend and geometry); every multiply / add is individually rounded (contraction off).
#include <math.h>

#include "hdn_common.h"

#pragma clang fp contract(off)

namespace hdn {

// softmax(1)[:, 1] of a 2-class logit pair as ATen's vectorised CPU kernel forms it: exp(x - max) for both, times 1 / sum.
__device__ __forceinline__ float softmax2_class1(float x0, float x1) {
  const float m = fmaxf(x0, x1);
  const float e0 = expf(rn_sub(x0, m)), e1 = expf(rn_sub(x1, m));
  return rn_mul(e1, rn_div(1.0f, rn_add(e0, e1)));
}

// score.sigmoid() of a 1-channel logit map (hdnTracker._convert_score, hdn_tracker.py:85-87) as ATen's CPU kernel forms it: 1 / (1 + exp(0 - x)).
__device__ __forceinline__ float sigmoid1(float x) { return rn_div(1.0f, rn_add(1.0f, expf(rn_sub(0.0f, x)))); }

// _convert_score of anchor i: cls_out_channels == 2 -> softmax class 1 of planes (0, 1); == 1 -> sigmoid of the single plane
__device__ __forceinline__ float class_score(const float* __restrict__ cls, int n, int i, int cls_channels) {
  return cls_channels == 1 ? sigmoid1(cls[i]) : softmax2_class1(cls[i], cls[n + i]);
}

// np.argmax: the largest value, the lowest index among equals.
__device__ __forceinline__ void wave_argmax(double& v, int& i) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const double ov = __shfl_xor(v, s);
    const int oi = __shfl_xor(i, s);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

__global__ __launch_bounds__(HDN_WAVE) void similarity_translation_kernel(const float* __restrict__ cls, const float* __restrict__ loc_c,
                                                                          const double* __restrict__ window, const float* __restrict__ points,
                                                                          const double* __restrict__ seq, double* __restrict__ state, int S,
                                                                          double window_influence, float stride_c, double exemplar,
                                                                          int cls_channels) {
  const int b = blockIdx.x, lane = threadIdx.x, n = S * S;
  cls += size_t(b) * cls_channels * n;
  loc_c += size_t(b) * 2 * n;
  seq += size_t(b) * HDN_SIM_SEQ_DOUBLES;
  state += size_t(b) * HDN_SIM_STATE_DOUBLES;
  const float keep = (float)(1.0 - window_influence);  // score (float32 array) * python float stays float32
  double best = -INFINITY;
  int best_i = 0x7fffffff;
  for (int i = lane; i < n; i += HDN_WAVE) {
    const float sc = class_score(cls, n, i, cls_channels);
    const double ps = (double)rn_mul(sc, keep) + window[i] * window_influence;
    if (ps > best) { best = ps; best_i = i; }  // ascending i per lane: the first maximum stays
  }
  wave_argmax(best, best_i);
  if (lane == 0) {
    const double scale_z = exemplar / seq[2];
    double dcx = 0.0, dcy = 0.0, stop = 0.0;
    if (best < 0.05) {
      stop = 1.0;
    } else {
      const float px = rn_sub(points[2 * best_i], rn_mul(loc_c[best_i], stride_c));
      const float py = rn_sub(points[2 * best_i + 1], rn_mul(loc_c[n + best_i], stride_c));
      dcx = (double)px / scale_z;
      dcy = (double)py / scale_z;
    }
    const double cx = dcx + seq[0], cy = dcy + seq[1];
    state[0] = dcx; state[1] = dcy; state[2] = cx; state[3] = cy; state[4] = stop;
    state[5] = (double)class_score(cls, n, best_i, cls_channels);  // best_score = score[best_idx] (:205)
    state[6] = (double)best_i; state[7] = best;
    // parameters of the second search crop, get_subwindow(img, self.center_pos, INSTANCE_SIZE, s_x, avg) (:191-193)
    state[8] = cx; state[9] = cy; state[10] = seq[3]; state[11] = seq[5]; state[12] = seq[6]; state[13] = seq[7];
  }
}

__device__ __forceinline__ void mat3_mul(const double* a, const double* b, double* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[3 * r + c] = (a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c]) + a[3 * r + 2] * b[6 + c];
}

__global__ __launch_bounds__(HDN_WAVE) void similarity_logpolar_kernel(const float* __restrict__ cls_lp, const float* __restrict__ loc_lp,
                                                                       const float* __restrict__ points_lp, const double* __restrict__ seq,
                                                                       double* __restrict__ state, int S, float stride_lp, double mag,
                                                                       float rot_unit, int cls_channels) {
  const int b = blockIdx.x, lane = threadIdx.x, n = S * S;
  cls_lp += size_t(b) * cls_channels * n;
  loc_lp += size_t(b) * 4 * n;
  seq += size_t(b) * HDN_SIM_SEQ_DOUBLES;
  state += size_t(b) * HDN_SIM_STATE_DOUBLES;
  double best = -INFINITY;  // (float32 scores compared as doubles: exact)
  int best_i = 0x7fffffff;
  for (int i = lane; i < n; i += HDN_WAVE) {
    const double sc = (double)class_score(cls_lp, n, i, cls_channels);
    if (sc > best) { best = sc; best_i = i; }
  }
  wave_argmax(best, best_i);
  if (lane == 0) {
    const double dcx = state[0], dcy = state[1], cx = state[2], cy = state[3];
    double scale_delta = 1.0, rot_delta = 0.0;  // sim_lp = [1, 1, 0, 0]: 1 * cur_sz / init_s_z == 1 exactly
    double sim0 = 1.0, gated = 1.0;             // what hdnTracker.track_new (hdn_tracker.py:245-248) reads: sim_lp[0] itself and whether it is the gate's [1, 1, 0, 0]
    if (!(state[4] != 0.0 || best < 0.25)) {
      const float d0 = rn_sub(points_lp[2 * best_i], rn_mul(loc_lp[best_i], stride_lp));
      const float d2 = rn_sub(points_lp[2 * best_i + 1], rn_mul(loc_lp[2 * n + best_i], stride_lp));
      const float sc = (float)exp((double)d0 * mag);  // np.exp(float32 * np.float64) stored into the float32 array
      scale_delta = ((double)sc * seq[2]) / seq[2];   // sim_lp[0] * cur_sz / self.init_s_z, cur_sz = init_s_z (:158,206)
      rot_delta = (double)rn_mul(d2, rot_unit);
      sim0 = (double)sc;
      gated = 0.0;
    }
    // rot_scale_around_center_shift_tran(cx, cy, rot_delta, scale_delta, delta_cx, delta_cy)
    double tran[9] = {1, 0, dcx, 0, 1, dcy, 0, 0, 1}, tmp[9];
    if (fabs(scale_delta) > 0 && scale_delta != 1.0) {
      const double ms[9] = {scale_delta, 0, cx * (1 - scale_delta), 0, scale_delta, cy * (1 - scale_delta), 0, 0, 1};
      mat3_mul(ms, tran, tmp);
#pragma unroll
      for (int q = 0; q < 9; ++q) tran[q] = tmp[q];
    }
    if (fabs(rot_delta) > 0) {
      const double cc = cos(rot_delta), ss = sin(rot_delta);
      const double mr[9] = {cc, -ss, (cx - cx * cc) + cy * ss, ss, cc, (cy - cy * cc) - cx * ss, 0, 0, 1};
      mat3_mul(mr, tran, tmp);
#pragma unroll
      for (int q = 0; q < 9; ++q) tran[q] = tmp[q];
    }
    state[16] = scale_delta; state[17] = rot_delta; state[18] = (double)best_i; state[19] = best;
#pragma unroll
    for (int q = 0; q < 9; ++q) state[20 + q] = tran[q];
    // img_rot_around_center(img, cx, cy, w, h, -rot_delta): the 2x3 matrix handed to cv2.warpAffine (transform.py:81-97)
    const double cc = cos(-rot_delta), ss = sin(-rot_delta);
    state[32] = cc; state[33] = -ss; state[34] = (cx - cx * cc) + cy * ss;
    state[35] = ss; state[36] = cc;  state[37] = (cy - cy * cc) - cx * ss;
    // get_subwindow_for_homo(rot_img_homo, self.center_pos, EXEMPLAR_SIZE, self.init_s_z_sm * scale_delta, avg) (:224-227)
    state[40] = cx; state[41] = cy; state[42] = seq[4] * scale_delta; state[43] = seq[5]; state[44] = seq[6]; state[45] = seq[7];
    state[46] = sim0; state[47] = gated;
  }
}

// ---- hdnTracker.track_new after the two decodes (hdn/tracker/hdn_tracker.py:213-301): TRACKS['hdnTracker'], the similarity-only tracker ----
// One lane per sequence, float64 as numpy computes these lines — except the two accumulators numpy itself turns into float32:
// `self.rot += sim_lp[2]` and `self.lp_shift[1] += sim_lp[2]` add an np.float32 to a python number, which NumPy 2 evaluates (and keeps) in
// float32 from the first un-gated frame on; a gated frame adds the python int 0 and changes neither value nor type.  tr[11] / tr[12] carry that type.
__device__ __forceinline__ void acc_f32(double& v, double& is_f32, float x) {
  v = (double)rn_add((float)v, x);
  is_f32 = 1.0;
}

__device__ __forceinline__ void wrap_2pi(double& v, double is_f32, double sign) {   // v -= sign * (math.pi * 2) in v's own type
  const double two_pi = 6.283185307179586;
  v = is_f32 != 0.0 ? (double)rn_sub((float)v, (float)(sign * two_pi)) : v - sign * two_pi;
}

__global__ __launch_bounds__(HDN_WAVE) void simi_track_update_kernel(const double* __restrict__ state, double* __restrict__ tr,
                                                                     double* __restrict__ seq, double* __restrict__ out, int B, double img_w,
                                                                     double img_h, double scale_score_thresh, double context_amount,
                                                                     double ratio) {
  const int b = blockIdx.x * HDN_WAVE + threadIdx.x;
  if (b >= B) return;
  const double* S = state + size_t(b) * HDN_SIM_STATE_DOUBLES;
  double* T = tr + size_t(b) * HDN_SIMI_TRACK_DOUBLES;
  double* O = out + size_t(b) * HDN_SIMI_OUT_DOUBLES;
  const double dcx = S[0], dcy = S[1], cx = S[2], cy = S[3], best_score = S[5], pscore = S[7];
  // :214-222  the "lost" bookkeeping (its only product, the next window_scale_factor, is overwritten by `= 1` at :183 before it is read)
  double new_wsf = 1.0, lost_count = T[9], last_lost = T[10];
  if (pscore < scale_score_thresh) {
    new_wsf = 1.5;
    if (lost_count == 0.0) last_lost = 1.0;
    lost_count += 1.0;
    if (last_lost == 0.0 && lost_count < 5.0) { lost_count = 0.0; last_lost = 0.0; }
  }
  // :224-227  (center = pred_c / scale_z * window_scale_factor with window_scale_factor = s_x / (s_z * ratio) == 1 exactly: s_z and ratio are
  //           integer-valued doubles, s_x = floor(s_z * ratio * 1) is their exact product — the translation decode's delta IS center)
  const double d = sqrt(dcx * dcx + dcy * dcy);
  T[7] = (T[13] == 0.0) ? d : (T[7] + d) / 2.0;      // fr_idx == 1  <=>  the first tracked frame of the sequence
  T[13] += 1.0;
  T[0] = cx; T[1] = cy;                                // :229-233
  // :247-252  size update, clamped
  const double sim0 = S[46];
  double width = T[2] * sim0, height = T[3] * sim0;    // (* window_scale_factor == 1.0; sim_lp[1] is the same float32 as sim_lp[0])
  const double lo = (10.0 * T[14]) / T[15];
  double m = (img_w < width) ? img_w : width;          // python min(width, W) / max(lo, .): the SECOND argument only if it is strictly smaller / larger
  width = (m > lo) ? m : lo;
  m = (img_h < height) ? img_h : height;
  height = (m > 10.0) ? m : 10.0;
  T[2] = width; T[3] = height;
  // :256-258
  double rot = T[4], lp = T[5], rot_f32 = T[11], lp_f32 = T[12];
  if (S[47] == 0.0) {
    const float rd = (float)S[17];
    acc_f32(lp, lp_f32, rd);
    acc_f32(rot, rot_f32, rd);
  }
  T[6] = width / T[14];
  O[0] = cx - width / 2.0; O[1] = cy - height / 2.0; O[2] = width; O[3] = height;
  // :264-269  (a float32 rot is compared with the python float rounded to float32, as NumPy 2 compares a float32 scalar with a weak python float)
  const double two_pi = 6.283185307179586;
  const double hi = rot_f32 != 0.0 ? (double)(float)two_pi : two_pi, lw = rot_f32 != 0.0 ? (double)(float)(-two_pi) : -two_pi;
  if (rot >= hi) { wrap_2pi(rot, rot_f32, 1.0); wrap_2pi(lp, lp_f32, 1.0); }
  else if (rot < lw) { wrap_2pi(rot, rot_f32, -1.0); wrap_2pi(lp, lp_f32, -1.0); }
  T[4] = rot; T[5] = lp; T[11] = rot_f32; T[12] = lp_f32;
  // :271-280  polygon = roll(transformPoly(cetner2poly([cx, cy, w, h]), getRotMatrix(cx, cy, rot)), 4 - poly_shift_l); np.cos / np.sin in rot's dtype
  const double cc = rot_f32 != 0.0 ? (double)(float)cos(rot) : cos(rot), ss = rot_f32 != 0.0 ? (double)(float)sin(rot) : sin(rot);
  const double tx = (cx - cx * cc) + cy * ss, ty = (cy - cy * cc) - cx * ss;
  const double x1 = cx - width * 0.5, y1 = cy - height * 0.5, x2 = cx + width * 0.5, y2 = cy + height * 0.5;
  const double px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
  const int shift = (4 - (int)T[19]) & 3;
  double mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double qx = (px[i] * cc + py[i] * (-ss)) + tx, qy = (px[i] * ss + py[i] * cc) + ty;
    const int j = (i + shift) & 3;
    O[10 + 2 * j] = qx; O[11 + 2 * j] = qy;
    mnx = fmin(mnx, qx); mny = fmin(mny, qy); mxx = fmax(mxx, qx); mxy = fmax(mxy, qy);
  }
  O[4] = mnx; O[5] = mny; O[6] = mxx - mnx; O[7] = mxy - mny;
  O[8] = best_score; O[9] = rot; O[18] = S[4]; O[19] = pscore;
  // :156-162  update_template: img_rot_around_center(init_img, init_pos, lp_shift[1]) (math.cos / math.sin: double, transform.py:80-97)
  const double ix = T[17], iy = T[18], c2 = cos(lp), s2 = sin(lp);
  T[32] = c2; T[33] = -s2; T[34] = (ix - ix * c2) + iy * s2;
  T[35] = s2; T[36] = c2;  T[37] = (iy - iy * c2) - ix * s2;
  // :283  window_scale_factor = new_window_scale_factor; then the next frame's :180-192: s_z, s_x and the first search crop
  T[8] = new_wsf; T[9] = lost_count; T[10] = last_lost;
  const double sum = width + height;
  const double w_z = width + context_amount * sum, h_z = height + context_amount * sum;
  const double s_z = floor(sqrt(w_z * h_z)), s_x = floor(s_z * ratio);
  double* q = seq + size_t(b) * HDN_SIM_SEQ_DOUBLES;
  q[0] = cx; q[1] = cy; q[2] = s_z; q[3] = s_x; q[4] = 0.0; q[5] = T[20]; q[6] = T[21]; q[7] = T[22];
  T[24] = cx; T[25] = cy; T[26] = s_x; T[27] = T[20]; T[28] = T[21]; T[29] = T[22];
}

// ---- the tracker's 3x3 bookkeeping (hdn_tracker_proj_e2e.py:150-155 and :251-272), one lane per sequence ----------------------

__device__ __forceinline__ double det3(const double* m) {
  return (m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6])) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

__global__ __launch_bounds__(HDN_WAVE) void track_prepare_kernel(const double* __restrict__ H_total, double* __restrict__ Ht,
                                                                 double* __restrict__ Hinv, int B) {
  const int b = blockIdx.x * HDN_WAVE + threadIdx.x;
  if (b >= B) return;
  double m[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) m[q] = H_total[9 * b + q];
  double det = det3(m);
  if (det == 0.0) {   // :150-153  "we will do inv after, so make sure H_total is non-singular"
#pragma unroll
    for (int q = 0; q < 9; ++q) m[q] = (q % 4 == 0) ? 1.0 : 0.0;
    det = 1.0;
  }
  const double adj[9] = {m[4] * m[8] - m[5] * m[7], m[2] * m[7] - m[1] * m[8], m[1] * m[5] - m[2] * m[4],
                         m[5] * m[6] - m[3] * m[8], m[0] * m[8] - m[2] * m[6], m[2] * m[3] - m[0] * m[5],
                         m[3] * m[7] - m[4] * m[6], m[1] * m[6] - m[0] * m[7], m[0] * m[4] - m[1] * m[3]};
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    Ht[9 * b + q] = m[q];
    Hinv[9 * b + q] = adj[q] / det;
  }
}

__global__ __launch_bounds__(HDN_WAVE) void track_accumulate_kernel(const double* __restrict__ Ht, const double* __restrict__ sim_state,
                                                                    const double* __restrict__ H_comp, const float* __restrict__ homo_score,
                                                                    const double* __restrict__ consts, const double* __restrict__ init_points,
                                                                    int n_points, double* __restrict__ H_out, float* __restrict__ out, int B) {
  const int b = blockIdx.x * HDN_WAVE + threadIdx.x;
  if (b >= B) return;
  const double* k = consts + size_t(b) * HDN_TRACK_CONST_DOUBLES;
  double t0[9], t1[9], H[9];
  // :251-258  H_homo = inv(shift_H) @ (inv(scale_H_1) @ H_hm_comp @ scale_H_1) @ shift_H, left to right as numpy evaluates it
  mat3_mul(k, H_comp + 9 * b, t0);
  mat3_mul(t0, k + 9, t1);
  mat3_mul(k + 18, t1, t0);
  mat3_mul(t0, k + 27, t1);                       // t1 = H_homo
  // :261-264  H = H_total @ H_sim  (@ H_homo unless homo_score > gate)
  if (sim_state) {
    mat3_mul(Ht + 9 * b, sim_state + size_t(b) * HDN_SIM_STATE_DOUBLES + 20, t0);
  } else {
#pragma unroll
    for (int q = 0; q < 9; ++q) t0[q] = Ht[9 * b + q];
  }
  if ((double)homo_score[b] > k[36]) {
#pragma unroll
    for (int q = 0; q < 9; ++q) H[q] = t0[q];
  } else {
    mat3_mul(t0, t1, H);
  }
  const double inv22 = 1.0 / H[8];                // :265  H = (1.0 / H.item(8)) * H
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    H[q] = inv22 * H[q];
    H_out[9 * b + q] = H[q];
  }
  // :272  cv2.perspectiveTransform(init_points (float32), H_total): double arithmetic, w = 1 / w, zero output for w == 0
  float* o = out + size_t(b) * (2 * n_points + 1);
  for (int i = 0; i < n_points; ++i) {
    const double x = (double)(float)init_points[(size_t(b) * n_points + i) * 2], y = (double)(float)init_points[(size_t(b) * n_points + i) * 2 + 1];
    double w = (x * H[6] + y * H[7]) + H[8];
    if (fabs(w) > 2.220446049250313e-16) {
      w = 1.0 / w;
      o[2 * i] = (float)(((x * H[0] + y * H[1]) + H[2]) * w);
      o[2 * i + 1] = (float)(((x * H[3] + y * H[4]) + H[5]) * w);
    } else {
      o[2 * i] = o[2 * i + 1] = 0.f;
    }
  }
  o[2 * n_points] = sim_state ? (float)sim_state[size_t(b) * HDN_SIM_STATE_DOUBLES + 5] : 0.f;   // best_score, as track_new returns it
}

}  // namespace hdn

extern "C" int hdn_similarity_translation_f32(const float* cls, const float* loc_c, const double* window, const float* points,
                                              const double* seq, double* state, int B, int S, double window_influence, float stride_c,
                                              double exemplar_size, int cls_channels, void* stream) {
  if (!cls || !loc_c || !window || !points || !seq || !state) return HDN_E_NULL;
  if (B <= 0 || S <= 0 || !(exemplar_size > 0) || (cls_channels != 1 && cls_channels != 2)) return HDN_E_SHAPE;
  if (S > 1024) return HDN_E_LIMIT;
  hipLaunchKernelGGL(hdn::similarity_translation_kernel, dim3(B), dim3(HDN_WAVE), 0, (hipStream_t)stream, cls, loc_c, window, points, seq, state,
                     S, window_influence, stride_c, exemplar_size, cls_channels);
  return hdn::launch_status();
}

extern "C" int hdn_similarity_logpolar_f32(const float* cls_lp, const float* loc_lp, const float* points_lp, const double* seq, double* state,
                                           int B, int S, float stride_lp, double mag, float rot_unit, int cls_channels, void* stream) {
  if (!cls_lp || !loc_lp || !points_lp || !seq || !state) return HDN_E_NULL;
  if (B <= 0 || S <= 0 || (cls_channels != 1 && cls_channels != 2)) return HDN_E_SHAPE;
  if (S > 1024) return HDN_E_LIMIT;
  hipLaunchKernelGGL(hdn::similarity_logpolar_kernel, dim3(B), dim3(HDN_WAVE), 0, (hipStream_t)stream, cls_lp, loc_lp, points_lp, seq, state, S,
                     stride_lp, mag, rot_unit, cls_channels);
  return hdn::launch_status();
}

extern "C" int hdn_simi_track_update_f64(const double* state, double* tr, double* seq, double* out, int B, int img_w, int img_h,
                                         double scale_score_thresh, double context_amount, double instance_exemplar_ratio, void* stream) {
  if (!state || !tr || !seq || !out) return HDN_E_NULL;
  if (B <= 0 || img_w <= 0 || img_h <= 0 || !(instance_exemplar_ratio > 0)) return HDN_E_SHAPE;
  hipLaunchKernelGGL(hdn::simi_track_update_kernel, dim3(hdn::cdiv(B, HDN_WAVE)), dim3(HDN_WAVE), 0, (hipStream_t)stream, state, tr, seq, out, B,
                     (double)img_w, (double)img_h, scale_score_thresh, context_amount, instance_exemplar_ratio);
  return hdn::launch_status();
}

extern "C" int hdn_track_prepare_f64(const double* H_total, double* Ht, double* Hinv, int B, void* stream) {
  if (!H_total || !Ht || !Hinv) return HDN_E_NULL;
  if (B <= 0) return HDN_E_SHAPE;
  if (Hinv == H_total || Hinv == Ht) return HDN_E_ALIAS;
  hipLaunchKernelGGL(hdn::track_prepare_kernel, dim3(hdn::cdiv(B, HDN_WAVE)), dim3(HDN_WAVE), 0, (hipStream_t)stream, H_total, Ht, Hinv, B);
  return hdn::launch_status();
}

extern "C" int hdn_track_accumulate_f64(const double* Ht, const double* sim_state, const double* H_comp, const float* homo_score,
                                        const double* consts, const double* init_points, int n_points, double* H_out, float* out, int B,
                                        void* stream) {
  if (!Ht || !H_comp || !homo_score || !consts || !init_points || !H_out || !out) return HDN_E_NULL;
  if (B <= 0 || n_points <= 0) return HDN_E_SHAPE;
  if (n_points > 4096) return HDN_E_LIMIT;
  hipLaunchKernelGGL(hdn::track_accumulate_kernel, dim3(hdn::cdiv(B, HDN_WAVE)), dim3(HDN_WAVE), 0, (hipStream_t)stream, Ht, sim_state, H_comp,
                     homo_score, consts, init_points, n_points, H_out, out, B);
  return hdn::launch_status();
}
