"""TEST INFRASTRUCTURE ONLY: CPU restatement (numpy + PyTorch-CPU) of the tracker loop hdn_amd.tracker.HomoTracker runs on
the device, i.e. of hdnTrackerHomo.init / track_new (hdn/tracker/hdn_tracker_proj_e2e.py:60-120,141-285) for the stages in
scope; the similarity estimate is the identity unless a `similarity` callable is given (SimilarityOracle below restates
hdn_tracker_proj_e2e.py:164-214 and is pinned to tests/golden/similarity.npz).  Built from oracle/frame_oracle.py (crops pinned, OpenCV pieces
restated and unpinned) and oracle/hdn_oracle.py (head pinned to the reference's goldens)."""
from __future__ import annotations

import numpy as np
import torch

from . import frame_oracle as F
from . import hdn_oracle as O


EXEMPLAR_SIZE, INSTANCE_SIZE, STRIDE, STRIDE_LP, BASE_SIZE, OUTPUT_SIZE_LP = 127, 255, 8, 8, 8, 13   # hdn/core/config.py
WINDOW_INFLUENCE = 0.1632532824922313   # experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml (config.py:530 default 0.45)
CONTEXT_AMOUNT = 0.5


def generate_points(stride, size):
    """hdnTracker.generate_points / generate_points_lp (hdn/tracker/hdn_tracker.py:32-49): cell centres, float32 [size*size, 2]."""
    ori = -(size // 2) * stride
    x, y = np.meshgrid([ori + stride * dx for dx in np.arange(0, size)], [ori + stride * dy for dy in np.arange(0, size)])
    points = np.zeros((size * size, 2), dtype=np.float32)
    points[:, 0], points[:, 1] = x.astype(np.float32).flatten(), y.astype(np.float32).flatten()
    return points


def hanning_window(score_size):
    """hdn_tracker_proj_e2e.py:26-29: np.outer(np.hanning(n), np.hanning(n)).flatten(), float64."""
    h = np.hanning(score_size)
    return np.outer(h, h).flatten()


def convert_score(score):
    """hdnTracker._convert_score (hdn_tracker.py:84-91): cls_out_channels = 2 (every shipped configuration) -> softmax over the two
    classes, class 1; cls_out_channels = 1 -> sigmoid of the single map (:85-87).  The channel count is the map's own."""
    ch = score.shape[1]
    if ch == 1:
        return score.permute(1, 2, 3, 0).contiguous().view(-1).sigmoid().detach().cpu().numpy()
    score = score.permute(1, 2, 3, 0).contiguous().view(ch, -1).permute(1, 0)
    return score.softmax(1).detach()[:, 1].cpu().numpy()


def convert_c(delta, point):
    """SiameseTracker._convert_c (base_tracker.py:54-59)."""
    delta = delta.permute(1, 2, 3, 0).contiguous().view(2, -1).detach().cpu().numpy().copy()
    delta[0, :] = point[:, 0] - delta[0, :] * 8
    delta[1, :] = point[:, 1] - delta[1, :] * 8
    return delta


def convert_logpolar_simi(delta, point):
    """hdnTracker._convert_logpolar_simi (hdn_tracker.py:51-67): rows (scale, scale, rotation, unused)."""
    delta = delta.permute(1, 2, 3, 0).contiguous().view(4, -1).detach().cpu().numpy().copy()
    delta[2, :] = point[:, 1] - delta[2, :] * STRIDE_LP
    delta[3, :] = point[:, 1] + delta[3, :] * STRIDE_LP
    delta[0, :] = point[:, 0] - delta[0, :] * STRIDE_LP
    delta[1, :] = point[:, 0] + delta[1, :] * STRIDE_LP
    scale = delta[0, :]
    rotation = delta[2, :]
    rotation = rotation * (2 * np.pi / EXEMPLAR_SIZE)
    mag = np.log(EXEMPLAR_SIZE / 2) / EXEMPLAR_SIZE
    delta[0, :] = np.exp(scale * mag)        # (float32 array * np.float64 scalar: float64 under NumPy 2, stored back as float32)
    delta[1, :] = delta[0, :]
    delta[2, :] = rotation
    return delta


def rot_scale_around_center_shift_tran(cx, cy, rot, scale, sx, sy):
    """hdn/utils/transform.py:250-298, float64."""
    import math
    tran = np.array([[1, 0, sx], [0, 1, sy], [0, 0, 1]]).astype(np.float64)
    if abs(scale) > 0 and scale != 1:
        tran = np.array([[scale, 0, cx * (1 - scale)], [0, scale, cy * (1 - scale)], [0, 0, 1]]).astype(np.float64) @ tran
    if abs(rot) > 0:
        cc, ss = math.cos(rot), math.sin(rot)
        tran = np.array([[cc, -ss, cx - cx * cc + cy * ss], [ss, cc, cy - cy * cc - cx * ss], [0, 0, 1]]).astype(np.float64) @ tran
    return tran


def decode_translation(cls, loc_c, window, points, init_s_z, window_influence=WINDOW_INFLUENCE):
    """hdn_tracker_proj_e2e.py:169-186: score, window blend, argmax, the 0.05 gate -> (center [2] in frame pixels, stop flag, ...)."""
    scale_z = EXEMPLAR_SIZE / np.float64(init_s_z)   # init_s_z is np.floor's np.float64 in the tracker (:97): float32 / it -> float64
    score = convert_score(cls)
    pred_c = convert_c(loc_c, points)
    pscore = score * (1 - window_influence) + window * window_influence
    best_idx = int(np.argmax(pscore))
    stop = 0
    if pscore[best_idx] < 0.05:
        center, stop = np.array([0.0, 0.0]), 1
    else:
        center = pred_c[:, best_idx] / scale_z
    return {"score": score, "pred_c": pred_c, "pscore": pscore, "best_idx": best_idx, "stop": stop,
            "center": np.asarray(center, np.float64), "best_score": score[best_idx]}


def decode_logpolar(cls_lp, loc_lp, points_lp, stop, cur_sz, init_s_z):
    """hdn_tracker_proj_e2e.py:197-212: argmax of the log-polar score, _convert_logpolar_simi, the 0.25 gate -> (scale_delta, rot_delta)."""
    score_lp = convert_score(cls_lp)
    pred = convert_logpolar_simi(loc_lp, points_lp)
    best = int(np.argmax(score_lp))
    sim_lp = pred[:, best]
    if stop or score_lp[best] < 0.25:
        sim_lp = [1, 1, 0, 0]
    scale_delta = sim_lp[0] * np.float64(cur_sz) / np.float64(init_s_z)
    rot_delta = sim_lp[2]
    return {"score_lp": score_lp, "pred_center_lp": pred, "best_idx_lp": best, "sim_lp": np.asarray(sim_lp, np.float64),
            "scale_delta": float(scale_delta), "rot_delta": float(rot_delta), "rot_delta_raw": rot_delta}


class SimilarityOracle:
    """The similarity half of track_new on the CPU (hdn_tracker_proj_e2e.py:157-214) around a model exposing the reference's
    ModelBuilder.track_new / track_new_lp (model_builder_e2e_unconstrained_v2.py:131-158); crops from oracle/frame_oracle.py.
    Call: (stabilised frame uint8 [H,W,3], init_pos, init_s_z, channel_average) -> dict(dcx, dcy, cx, cy, scale_delta,
    rot_delta, best_score, H_sim)."""

    def __init__(self, model, window_influence=WINDOW_INFLUENCE, instance_size=INSTANCE_SIZE):
        """instance_size: cfg.TRACK.INSTANCE_SIZE (255 shipped; 303 = BASELINE configs[4], score map 31 x 31, :24-25)."""
        self.model, self.window_influence, self.instance_size = model, window_influence, int(instance_size)
        score_size = (self.instance_size - EXEMPLAR_SIZE) // STRIDE + 1 + BASE_SIZE
        self.score_size = score_size
        self.window = hanning_window(score_size)
        self.points = generate_points(STRIDE, score_size)
        self.points_lp = generate_points(STRIDE_LP, OUTPUT_SIZE_LP)

    def __call__(self, img, init_pos, init_s_z, channel_average):
        s_x = np.floor(init_s_z * np.round(self.instance_size / EXEMPLAR_SIZE))
        x_crop = F.get_subwindow(img, init_pos, self.instance_size, s_x, channel_average)
        with torch.no_grad():
            out = self.model.track_new(torch.from_numpy(x_crop))
        tr = decode_translation(out["cls"], out["loc_c"], self.window, self.points, init_s_z, self.window_influence)
        cx, cy = tr["center"][0] + init_pos[0], tr["center"][1] + init_pos[1]
        x_moved = F.get_subwindow(img, np.array([cx, cy]), self.instance_size, s_x, channel_average)
        with torch.no_grad():
            out = self.model.track_new_lp(torch.from_numpy(x_moved), [0, 0])
        lp = decode_logpolar(out["cls_lp"], out["loc_lp"], self.points_lp, tr["stop"], init_s_z, init_s_z)
        H_sim = rot_scale_around_center_shift_tran(cx, cy, lp["rot_delta"], lp["scale_delta"], tr["center"][0], tr["center"][1])
        return {"dcx": tr["center"][0], "dcy": tr["center"][1], "cx": cx, "cy": cy, "scale_delta": lp["scale_delta"],
                "rot_delta": lp["rot_delta"], "best_score": float(tr["best_score"]), "H_sim": H_sim,
                "s_x": s_x, "x_crop": x_crop, "x_crop_moved": x_moved, "translation": tr, "logpolar": lp}


def track_prepare(H_total):
    """hdn_tracker_proj_e2e.py:150-155: a singular H_total is reset to the identity; -> (H_total, the matrix handed to cv2.warpPerspective)."""
    if np.linalg.det(H_total) == 0:
        H_total = np.eye(3, dtype=np.float32)
    return H_total, np.linalg.inv(H_total)


def perspective_transform(points, H):
    """cv2.perspectiveTransform (hdn_tracker_proj_e2e.py:272) restated — OpenCV's perspectiveTransform_<float> with the matrix
    converted to double: per point x*m0 + y*m1 + m2 etc. in double, multiplied by 1 / w (0 when |w| is below eps), stored as
    float32.  PARITY UNPINNED like every OpenCV piece (no cv2 in the reference tree or the image).  points float32 [N,2]."""
    H = np.asarray(H, np.float64).reshape(3, 3)
    q = np.asarray(points, np.float32).reshape(-1, 2).astype(np.float64)
    w = q[:, 0] * H[2, 0] + q[:, 1] * H[2, 1] + H[2, 2]
    ok = np.abs(w) > np.finfo(np.float64).eps
    w = np.where(ok, 1.0 / np.where(ok, w, 1.0), 0.0)
    pts = np.stack([(q[:, 0] * H[0, 0] + q[:, 1] * H[0, 1] + H[0, 2]) * w, (q[:, 0] * H[1, 0] + q[:, 1] * H[1, 1] + H[1, 2]) * w], 1)
    return pts.astype(np.float32)


def track_accumulate(H_total, H_sim, H_hm_comp, score, z_crop_points_sm, init_points, score_gate=2.5, return_homo=False):
    """hdn_tracker_proj_e2e.py:251-272: un-scale / un-shift the residual, gate it, accumulate, project the initial corners
    (cv2.perspectiveTransform restated: double arithmetic on the float32 points, multiplied by 1 / w)."""
    cw = z_crop_points_sm[2] - z_crop_points_sm[0] + 1
    ch = z_crop_points_sm[3] - z_crop_points_sm[1] + 1
    S = np.array([[127 / cw, 0, 0], [0, 127 / ch, 0], [0, 0, 1]]).astype(np.float32)
    H_hm_comp = np.linalg.inv(S) @ H_hm_comp @ S
    Sh = np.array([[1, 0, -z_crop_points_sm[0]], [0, 1, -z_crop_points_sm[1]], [0, 0, 1]]).astype(np.float32)
    H_homo = np.linalg.inv(Sh) @ H_hm_comp @ Sh
    H = H_total @ H_sim if float(np.asarray(score).reshape(-1)[0]) > score_gate else H_total @ H_sim @ H_homo
    H = (1.0 / H.item(8)) * H
    if return_homo:
        return H, perspective_transform(init_points, H), H_homo
    return H, perspective_transform(init_points, H)


def refine_loop(track_proj, init_homo_tmp, homo_search_img, iterations=1):
    """hdn_tracker_proj_e2e.py:241-250 as the reference executes it: per iteration ModelBuilder.track_proj on the (template, search)
    pair (float32 tensors made from the float64 crops, homo_estimate :42-57), H_hm = inv(H_mat) in float32 (np.linalg.inv of a float32
    array), normalised by its last element, the float64 search crop warped by cv2.warpPerspective(inv(H_hm)) for the next iteration,
    H_hm_comp (float64) @= H_hm.  `track_proj(template [1,1,127,127] float32, search [1,1,127,127] float32) -> (H_mat [1,3,3], score, simi)`.
    -> (H_hm_comp float64 [3,3], score of the last iteration, list of per-iteration (H_mat, H_hm))."""
    H_hm_comp = np.identity(3)
    tmpl = torch.Tensor(np.asarray(init_homo_tmp)).float().unsqueeze(0)
    cur = np.asarray(homo_search_img)
    steps, score = [], None
    for _ in range(iterations):
        with torch.no_grad():
            H_mat, score, _ = track_proj(tmpl, torch.Tensor(cur).float().unsqueeze(0))
        score = np.asarray(score.detach().cpu().numpy() if isinstance(score, torch.Tensor) else score)
        H_hm = H_mat.detach().cpu().squeeze(0).numpy()
        H_hm = np.linalg.inv(H_hm)
        H_hm = (1.0 / H_hm.item(8)) * H_hm
        cur = np.expand_dims(O.warp_perspective_replicate(cur[0], np.linalg.inv(H_hm).astype(np.float64)), 0)
        H_hm_comp = H_hm_comp @ H_hm
        steps.append((H_mat.detach().cpu().numpy().copy(), H_hm.copy()))
    return H_hm_comp, score, steps


class HomoTrackerOracle:
    """hdnTrackerHomo (hdn_tracker_proj_e2e.py:22-285) on the CPU.  The homography estimate is either `track_proj` — a callable with
    ModelBuilder.track_proj's role, see refine_loop — or, when that is None, the oracle's own track_proj (oracle/hdn_oracle.py) around
    PreShareFeature weights `sf_sd` and a trunk `regress`.  After init / every track_new `self.trace` holds the intermediates the
    reference's method has as local variables (crops, stabilised frame, H_sim, H_hm, ...): tests/test_oracle_golden.py holds them to
    tests/golden/tracker_loop.npz, the values the reference's own init / track_new produced when they were executed."""

    def __init__(self, sf_sd: dict, regress, iterations: int = 1, score_gate: float = 2.5, similarity: SimilarityOracle = None, track_proj=None):
        self.sf_sd, self.regress, self.iterations, self.score_gate = sf_sd, regress, iterations, score_gate
        self.similarity = similarity
        self.trace = {}
        if track_proj is None:
            h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32)

            def track_proj(tmpl, srch):
                imgs = torch.cat([tmpl, srch], dim=1)
                Hm, score, simi, _ = O.track_proj({"org_imgs": imgs, "input_tensors": imgs, "h4p": h4p}, self.sf_sd, self.regress)
                return Hm, score, simi
        self._track_proj = track_proj

    def init(self, img, bbox, poly, gt_points, first_point=None):
        self.init_pos = np.array([poly[0], poly[1]], np.float64)
        self.center_pos = self.init_pos.copy()
        self.size = np.array([poly[2], poly[3]], np.float64)
        w_z = self.size[0] + CONTEXT_AMOUNT * np.sum(self.size)
        h_z = self.size[1] + CONTEXT_AMOUNT * np.sum(self.size)
        self.init_s_z = np.floor(np.sqrt(w_z * h_z))
        self.init_s_z_sm = float(np.floor(np.sqrt(self.size[0] * self.size[1])))
        self.channel_average = np.mean(img, axis=(0, 1))
        crop, self.z_crop_points_sm = F.get_subwindow_for_homo(img, self.init_pos, 127, self.init_s_z_sm, self.channel_average)
        self.init_homo_tmp = F.search_info(crop[0])                      # float64 [1,127,127]
        sm_log = F.get_polar_img(crop[0].transpose(1, 2, 0).astype(np.uint8)).transpose(2, 0, 1)[None].astype(np.float32)
        self.trace = {"z_crop_sm": np.concatenate([crop, sm_log], axis=1)}     # :103-107 asks for islog=1 here too; only [:, 0:3] is used (:119)
        # :99-107  z_crop with the log-polar channels appended (islog=1), model.template(z_crop)
        z, self.z_crop_points = F.get_subwindow_for_homo(img, self.init_pos, 127, self.init_s_z, self.channel_average)
        z_u8 = z[0].transpose(1, 2, 0).astype(np.uint8)
        z_log = F.get_polar_img(z_u8).transpose(2, 0, 1)[None].astype(np.float32)
        self.z_crop = np.concatenate([z, z_log], axis=1)
        if self.similarity is not None:
            with torch.no_grad():
                self.similarity.model.template(torch.from_numpy(self.z_crop))
        self.init_points = np.asarray(gt_points, np.float32).reshape(-1, 2)
        self.H_total = np.eye(3, dtype=np.float32)
        self.scale, self.rot = 1, poly[4] if len(poly) > 4 else 0.0

    def track_new(self, fr_idx, img):
        self.H_total, H_inv = track_prepare(self.H_total)
        img = F.warp_perspective_u8(img, H_inv)
        cx, cy = self.init_pos
        H_sim, scale_delta, rot_delta, best_score, sim = np.eye(3), 1.0, 0.0, 0.0, None
        rot_img = img   # rot_delta = 0: img_rot_around_center is the bicubic identity
        if self.similarity is not None:
            sim = self.similarity(img, self.init_pos, self.init_s_z, self.channel_average)
            cx, cy, H_sim, scale_delta, rot_delta, best_score = sim["cx"], sim["cy"], sim["H_sim"], sim["scale_delta"], sim["rot_delta"], sim["best_score"]
            self.center_pos = np.array([cx, cy])
            self.rot += sim["logpolar"]["rot_delta_raw"]   # (:215; np.float32 unless gated: the sum is float32 from the first un-gated frame on)
            self.scale *= scale_delta
            rot_img = F.warp_affine_cubic_u8(img, F.rot_matrix_2x3(cx, cy, -rot_delta))   # :223
        crop, crop_points = F.get_subwindow_for_homo(rot_img, np.array([cx, cy]), 127, self.init_s_z_sm * scale_delta, self.channel_average)
        search = F.search_info(crop[0])
        H_comp, score, steps = refine_loop(self._track_proj, self.init_homo_tmp, search, self.iterations)
        H, pts, H_homo = track_accumulate(self.H_total, H_sim, H_comp, score, self.z_crop_points_sm, self.init_points, self.score_gate, return_homo=True)
        self.H_total = H
        self.trace = {"H_homo": H_homo, "img": img, "rot_img": rot_img, "x_crop_homo": crop, "crop_points": crop_points, "search": search, "H_hm": steps[-1][1],
                      "H_mat": steps[-1][0], "H_hm_comp": H_comp, "homo_score": score, "H_sim": H_sim, "scale_delta": scale_delta, "rot_delta": rot_delta}
        return {"points": pts, "polygon": pts, "score": float(np.asarray(score).reshape(-1)[0]), "best_score": best_score, "similarity": sim}


# ------------------------------------------------------------------------------------------------ TRACKS['hdnTracker'] (similarity only)
SCALE_SCORE_THRESH = 0.5    # cfg.TRACK.SCALE_SCORE_THRESH (hdn/core/config.py:527, experiments/siamban_r50_l234_pot/config.yaml:48)


def center2poly(c):
    """cetner2poly (hdn/utils/bbox.py:40-56): (cx, cy, w, h) -> (x1, y1, x2, y1, x2, y2, x1, y2)."""
    x, y, w, h = c[0], c[1], c[2], c[3]
    x1, y1, x2, y2 = x - w * 0.5, y - h * 0.5, x + w * 0.5, y + h * 0.5
    return np.array([x1, y1, x2, y1, x2, y2, x1, y2])


def get_rot_matrix(cx, cy, rot):
    """getRotMatrix (hdn/utils/bbox.py:58-75): np.cos / np.sin of `rot` in ITS dtype (float32 once the accumulated rotation is)."""
    cc, ss = np.cos(rot), np.sin(rot)
    return np.array([[cc, -ss, cx - cx * cc + cy * ss], [ss, cc, cy - cy * cc - cx * ss], [0, 0, 1]])


def transform_poly(polygon, m):
    """transformPoly (hdn/utils/bbox.py:77-90)."""
    polygon = polygon.reshape(-1, 2)
    out = np.ones([polygon.shape[0], 3])
    out[:, 0:2] = polygon
    out = out @ m.transpose(1, 0)
    return out[:, 0:2]


class SimiTrackerOracle:
    """TEST INFRASTRUCTURE: hdnTracker (hdn/tracker/hdn_tracker.py:18-301) on the CPU — the similarity-only tracker with a template
    refresh every frame, TRACKS['hdnTracker'] (hdn/tracker/tracker_builder.py:13), the default cfg.TRACK.TYPE (hdn/core/config.py:517).
    `model` exposes ModelBuilder.template / track_new / track_new_lp.  Statement by statement as the reference, with numpy's scalar types
    left to fall where they fall under NumPy 2 (e.g. `self.rot += sim_lp[2]` turns a python number into np.float32 on the first un-gated
    frame); tests/test_oracle_golden.py holds it to tests/golden/tracker_loop_simi.npz = the reference's own init / track_new executed."""

    def __init__(self, model, window_influence=WINDOW_INFLUENCE, instance_size=INSTANCE_SIZE, scale_score_thresh=SCALE_SCORE_THRESH):
        self.model, self.window_influence, self.instance_size, self.scale_score_thresh = model, window_influence, int(instance_size), scale_score_thresh
        self.score_size = (self.instance_size - EXEMPLAR_SIZE) // STRIDE + 1 + BASE_SIZE
        self.window = hanning_window(self.score_size)
        self.points = generate_points(STRIDE, self.score_size)
        self.points_lp = generate_points(STRIDE_LP, OUTPUT_SIZE_LP)
        self.trace = {}

    def _template(self, img, s_z):
        z = F.get_subwindow(img, self.init_pos, EXEMPLAR_SIZE, s_z, self.channel_average)
        z_log = F.get_polar_img(z[0].transpose(1, 2, 0).astype(np.uint8)).transpose(2, 0, 1)[None].astype(np.float32)
        z_crop = np.concatenate([z, z_log], axis=1)                       # islog=1 (base_tracker.py:119-126)
        with torch.no_grad():
            self.model.template(torch.from_numpy(z_crop))
        return z_crop

    def init(self, img, bbox, poly, first_point):
        """hdn_tracker.py:109-154."""
        self.center_pos = np.array([poly[0], poly[1]])
        self.init_rot = poly[4]
        self.rot = poly[4]
        polygon = transform_poly(center2poly(poly[:4]), get_rot_matrix(poly[0], poly[1], poly[4]))
        fir_dis = (polygon - first_point) ** 2
        self.poly_shift_l = np.argmin(fir_dis[:, 0] + fir_dis[:, 1])
        self.scale, self.lp_shift, self.v = 1, [0, 0], 0
        self.size = np.array([poly[2], poly[3]])
        self.align_size = np.array([bbox[2], bbox[3]])
        w_z = self.size[0] + CONTEXT_AMOUNT * np.sum(self.size)
        h_z = self.size[1] + CONTEXT_AMOUNT * np.sum(self.size)
        s_z = np.floor(np.sqrt(w_z * h_z))
        self.channel_average = np.mean(img, axis=(0, 1))
        self.init_pos = np.array([poly[0], poly[1]])
        self.init_img, self.init_size, self.init_s_z = img, self.size, s_z
        self.window_scale_factor, self.lost_count, self.last_lost = 1.0, 0, False
        self.trace = {"z_crop": self._template(img, s_z)}

    def update_template(self):
        """:156-162: the FIRST frame rotated about init_pos by the accumulated rotation, cropped again, model.template again."""
        # (img_rot_around_center takes math.cos / math.sin of the angle: double arithmetic on the float32 sum's value, transform.py:80-81)
        img = F.warp_affine_cubic_u8(self.init_img, F.rot_matrix_2x3(self.init_pos[0], self.init_pos[1], float(self.lp_shift[1])))
        return self._template(img, self.init_s_z), img

    def track_new(self, fr_idx, img, gt_box=None, gt_poly=None):
        """:174-301."""
        import math
        w_z = self.size[0] + CONTEXT_AMOUNT * np.sum(self.size)
        h_z = self.size[1] + CONTEXT_AMOUNT * np.sum(self.size)
        self.window_scale_factor = 1
        s_z = np.floor(np.sqrt(w_z * h_z))
        ratio = np.round(self.instance_size / EXEMPLAR_SIZE)
        scale_z = EXEMPLAR_SIZE / s_z
        s_x = np.floor(s_z * ratio * self.window_scale_factor)
        self.window_scale_factor = s_x / (s_z * ratio)
        x_crop = F.get_subwindow(img, self.center_pos, self.instance_size, s_x, self.channel_average)
        with torch.no_grad():
            out = self.model.track_new(torch.from_numpy(x_crop))
        score = convert_score(out["cls"])
        pred_c = convert_c(out["loc_c"], self.points)
        pscore = score * (1 - self.window_influence) + self.window * self.window_influence
        best_idx = np.argmax(pscore)
        stop = 0
        if pscore[best_idx] < 0.05:
            center, stop = [0, 0], 1
        else:
            center = pred_c[:, best_idx] / scale_z * self.window_scale_factor
        new_wsf = 1
        if pscore[best_idx] < self.scale_score_thresh:
            new_wsf = 1.5
            if self.lost_count == 0:
                self.last_lost = True
            self.lost_count += 1
            if not self.last_lost and self.lost_count < 5:
                self.lost_count, self.last_lost = 0, False
        d = math.sqrt(center[0] * center[0] + center[1] * center[1])
        self.v = d if fr_idx == 1 else (self.v + d) / 2
        cx, cy = center[0] + self.center_pos[0], center[1] + self.center_pos[1]
        self.center_pos = np.array([cx, cy])
        x_moved = F.get_subwindow(img, self.center_pos, self.instance_size, s_x, self.channel_average)
        with torch.no_grad():
            out = self.model.track_new_lp(torch.from_numpy(x_moved), [0, 0])
        score_lp = convert_score(out["cls_lp"])
        pred_lp = convert_logpolar_simi(out["loc_lp"], self.points_lp)
        best_idx_lp = np.argmax(score_lp)
        sim_lp = pred_lp[:, best_idx_lp]
        if stop or score_lp[best_idx_lp] < 0.25:
            sim_lp = [1, 1, 0, 0]
        width = self.size[0] * sim_lp[0] * self.window_scale_factor
        height = self.size[1] * sim_lp[1] * self.window_scale_factor
        width = max(10 * self.init_size[0] / self.init_size[1], min(width, img.shape[:2][1]))
        height = max(10, min(height, img.shape[:2][0]))
        self.size = np.array([width, height])
        self.lp_shift[1] += sim_lp[2]
        self.rot += sim_lp[2]
        self.scale = width / self.init_size[0]
        bbox = [cx - width / 2, cy - height / 2, width, height]
        if self.rot >= 2 * math.pi:
            self.rot -= math.pi * 2
            self.lp_shift[1] -= math.pi * 2
        elif self.rot < -2 * math.pi:
            self.rot += math.pi * 2
            self.lp_shift[1] += math.pi * 2
        best_score = score[best_idx]
        polygon = transform_poly(center2poly([cx, cy, width, height]), get_rot_matrix(cx, cy, self.rot))
        polygon = np.roll(polygon, 4 - self.poly_shift_l, 0)
        max_p, min_p = np.max(polygon, 0), np.min(polygon, 0)
        align_bbox = [min_p[0], min_p[1], max_p[0] - min_p[0], max_p[1] - min_p[1]]
        self.align_size = [align_bbox[2], align_bbox[3]]
        z_crop, rot_img = self.update_template()
        self.window_scale_factor = new_wsf
        self.trace = {"s_x": s_x, "s_z": s_z, "x_crop": x_crop, "x_crop_moved": x_moved, "best_idx": int(best_idx), "stop": stop,
                      "pscore_best": pscore[best_idx], "center": np.asarray(center, np.float64), "best_idx_lp": int(best_idx_lp),
                      "score_lp_best": score_lp[best_idx_lp], "sim_lp": np.asarray(sim_lp, np.float64), "z_crop": z_crop, "rot_init_img": rot_img}
        return {"bbox": bbox, "bbox_aligned": align_bbox, "best_score": best_score, "rot": self.rot, "polygon": polygon}
