"""The per-frame tracker loop around the homography head, device-resident (BASELINE config 4).

    HomoTracker.init(img, bbox, poly, gt_points, first_point)        <- hdnTrackerHomo.init,      hdn/tracker/hdn_tracker_proj_e2e.py:60-120
    HomoTracker.track_new(fr_idx, img, gt_box, gt_poly, gt_points)   <- hdnTrackerHomo.track_new, hdn_tracker_proj_e2e.py:141-285

Same call signatures and the same result dictionary ('bbox_aligned', 'best_score', 'polygon', 'points', 'bbox') as the
reference's tracker, for the part of a frame this repository owns: undo the accumulated motion (full-frame warp by
inv(H_total), :154), cut the 127-px homography crop (get_subwindow_for_homo + get_search_info, :223-239), run the refinement
loop around ModelBuilder.track_proj (:242-250), un-scale / un-shift the residual (:251-258), gate it (`homo_score > 2.5`,
:261-264), accumulate H_total and project the initial corners (:266-272).

The similarity branch (translation and log-polar scale / rotation: ResNet-50 backbone, necks, MultiBAN / MultiCircBAN heads,
:164-214) belongs to PyTorch-ROCm and the reference's own modules (north_star); it enters here through the optional
`similarity` callable, `similarity(frame_u8_device, center_pos) -> (delta_cx, delta_cy, scale_delta, rot_delta, best_score)`.
Without one the similarity estimate is the identity: that is the harness configuration of tests/ and tests/tools/sequence_bench.py
(one planar target per sequence; sequences are independent, so N GPUs run N sequences: replicas only, no collective).

What the reference does on the host per frame — cv2 warps of the full frame, numpy crops, three crop uploads and six
.cpu().numpy() syncs — is here: ONE upload of the uint8 frame, kernels (hdn_amd.frame, hdn_amd.refine) and 3x3 float64
bookkeeping on the device; the only host read is the 4 projected corners the caller asks for (`sync=True`).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import frame as FR
from .refine import homo_refine

EXEMPLAR_SIZE = 127      # cfg.TRACK.EXEMPLAR_SIZE (hdn/core/config.py)
CONTEXT_AMOUNT = 0.5     # cfg.TRACK.CONTEXT_AMOUNT


def rot_scale_around_center_shift_tran(cx, cy, rot, scale, sx, sy):
    """hdn/utils/transform.py:250-298 (host, float64)."""
    tran = np.array([[1, 0, sx], [0, 1, sy], [0, 0, 1]], np.float64)
    if abs(scale) > 0 and scale != 1:
        tran = np.array([[scale, 0, cx * (1 - scale)], [0, scale, cy * (1 - scale)], [0, 0, 1]], np.float64) @ tran
    if abs(rot) > 0:
        cc, ss = math.cos(rot), math.sin(rot)
        tran = np.array([[cc, -ss, cx - cx * cc + cy * ss], [ss, cc, cy - cy * cc - cx * ss], [0, 0, 1]], np.float64) @ tran
    return tran


class HomoTracker:
    def __init__(self, hm_net, iterations: int = 1, similarity=None, score_gate: float = 2.5, graph: bool = False):
        """hm_net: hdn_amd.HomoModelBuilder (or the reference's, after install()) in eval mode on the GPU.
        iterations: trip count of the refinement loop (1 in the shipped tracker, :242; 2 in BASELINE config 5).
        graph: replay the whole per-frame body as ONE hipGraph (about 200 launches at B = 1 are launch-latency bound);
        only without a `similarity` callable, whose host-side decisions cannot be captured."""
        self.net = hm_net
        self.use_graph = bool(graph)
        self._graph = None
        self._capturing = False
        self.iterations = int(iterations)
        self.similarity = similarity
        self.score_gate = float(score_gate)
        self.host_syncs = 0  # device->host reads this object has caused (the reference: six per frame)

    # -------------------------------------------------------------------------------------------------- init
    def init(self, img, bbox, poly, gt_points, first_point=None):
        """img: BGR uint8 [H,W,3]; bbox (x, y, w, h); poly (cx, cy, w, h, theta); gt_points: the 4 corners."""
        self.dev = next(self.net.parameters()).device
        self.init_pos = np.array([poly[0], poly[1]], np.float64)
        self.center_pos = self.init_pos.copy()
        self.size = np.array([poly[2], poly[3]], np.float64)
        w_z = self.size[0] + CONTEXT_AMOUNT * np.sum(self.size)
        h_z = self.size[1] + CONTEXT_AMOUNT * np.sum(self.size)
        self.init_s_z = float(np.floor(np.sqrt(w_z * h_z)))
        self.init_s_z_sm = float(np.floor(np.sqrt(self.size[0] * self.size[1])))
        frame = FR.upload(img)
        # np.mean(img, axis=(0, 1)) of the first frame: one reduction on the device, read once per sequence
        self.channel_average = frame.to(torch.float64).mean(dim=(0, 1)).cpu().numpy()
        self.host_syncs += 1
        H, W, _ = frame.shape
        self.z_crop_points_sm = FR.crop_points(self.center_pos, self.init_s_z_sm, H, W)
        # get_template_info(get_subwindow_for_homo(...)[:, 0:3]) : the normalised gray template, constant for the sequence
        self.init_homo_tmp = FR.get_search_info(frame, self.center_pos, self.init_s_z_sm, self.channel_average)
        self.init_points = torch.tensor(np.asarray(gt_points, np.float64).reshape(-1, 2), dtype=torch.float64, device=self.dev)
        self._init_points_h = torch.cat([self.init_points, torch.ones_like(self.init_points[:, :1])], dim=1)
        self.H_total = torch.eye(3, dtype=torch.float64, device=self.dev)
        self._eye = torch.eye(3, dtype=torch.float64, device=self.dev)
        self._const_params = FR._dev_f64([self.init_pos[0], self.init_pos[1], self.init_s_z_sm] + [float(a) for a in self.channel_average], self.dev)
        self._graph = None
        # un-scale / un-shift of the residual (:251-258) are constants of the sequence: H_homo = A @ H_hm_comp @ B
        cw = self.z_crop_points_sm[2] - self.z_crop_points_sm[0] + 1
        ch = self.z_crop_points_sm[3] - self.z_crop_points_sm[1] + 1
        S = np.diag([EXEMPLAR_SIZE / cw, EXEMPLAR_SIZE / ch, 1.0]).astype(np.float32)          # float32, as :251-257 build them
        Sh = np.array([[1, 0, -self.z_crop_points_sm[0]], [0, 1, -self.z_crop_points_sm[1]], [0, 0, 1]], np.float32)
        A = np.linalg.inv(Sh).astype(np.float64) @ np.linalg.inv(S).astype(np.float64)           # (float32 inverses, numpy's dtype rule)
        self._A = torch.tensor(A, dtype=torch.float64, device=self.dev)
        self._B = torch.tensor(S.astype(np.float64) @ Sh.astype(np.float64), dtype=torch.float64, device=self.dev)

    # -------------------------------------------------------------------------------------------------- one frame
    @staticmethod
    def _det3(m):
        return (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
                + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))

    def _body(self, frame, params, H_sim, rot_delta=0.0, cx=None, cy=None):
        """Everything of a frame that runs on the device, from the uploaded frame to the 4 projected corners; no host
        reads, no allocations that depend on data: capturable in a hipGraph when its inputs are static buffers."""
        # :150-155  undo the accumulated motion (a singular H_total is reset to the identity, as the reference does).
        # hdn_frame_warp_perspective_u8 inverts its matrix itself (cv2 semantics), so it is handed inv(H_total) = adj / det.
        det = self._det3(self.H_total)
        Ht = torch.where(det == 0, self._eye, self.H_total)
        frame = FR.warp_perspective(frame, torch.linalg.inv(Ht).reshape(-1) if not self._capturing else self._inv3(Ht).reshape(-1))
        # :223-239  rotate back, cut the homography crop, normalise
        rot_img = FR.rot_around_center(frame, cx, cy, -rot_delta) if rot_delta != 0 else frame  # (rot 0: bicubic identity)
        search = FR.get_search_info(rot_img, None, None, None, params=params)
        # :242-250  refinement loop around track_proj
        H_comp, homo_score, _ = homo_refine(self.net, self.init_homo_tmp, search, iterations=self.iterations)
        # :251-266  un-scale, un-shift, gate, accumulate
        H_homo = self._A @ H_comp[0] @ self._B
        base = Ht @ H_sim
        H = torch.where(homo_score.to(torch.float64) > self.score_gate, base, base @ H_homo)
        H = H / H[2, 2]
        # :272  cv2.perspectiveTransform(init_points, H_total)
        p = self._init_points_h @ H.T
        pts = (p[:, :2] / p[:, 2:3]).to(torch.float32)
        return H, pts, homo_score

    def _inv3(self, m):
        """Closed-form 3x3 inverse from elementwise ops (graph capture cannot hold the solver call of torch.linalg.inv)."""
        a, b, c, d, e, f, g, h, i = (m[0, 0], m[0, 1], m[0, 2], m[1, 0], m[1, 1], m[1, 2], m[2, 0], m[2, 1], m[2, 2])
        adj = torch.stack([torch.stack([e * i - f * h, c * h - b * i, b * f - c * e]),
                           torch.stack([f * g - d * i, a * i - c * g, c * d - a * f]),
                           torch.stack([d * h - e * g, b * g - a * h, a * e - b * d])])
        return adj / self._det3(m)

    def _capture(self, frame_shape):
        """hipGraph of the whole per-frame body (similarity = identity only: its parameters are constants of the sequence).
        The frame lands in a static device buffer; H_total is carried in a static tensor updated by the graph itself."""
        self._static_frame = torch.empty(frame_shape, dtype=torch.uint8, device=self.dev)
        self._capturing = True
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        H0 = self.H_total.clone()
        with torch.cuda.stream(side):
            for _ in range(3):  # warm-up on the side stream (MIOpen find, lazy initialisations) without touching the state
                self._body(self._static_frame, self._const_params, self._eye)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            H, pts, score = self._body(self._static_frame, self._const_params, self._eye)
            self.H_total.copy_(H)            # the recurrence lives inside the graph
            self._g_pts, self._g_score = pts, score
        self.H_total.copy_(H0)
        self._capturing = False

    def track_new(self, fr_idx, img, gt_box=None, gt_poly=None, gt_points=None, sync: bool = True):
        best_score = 0.0
        if self.use_graph and self.similarity is None:
            t = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
            if self._graph is None:
                self._capture(tuple(t.shape))
            if tuple(t.shape) != tuple(self._static_frame.shape) or t.dtype != torch.uint8:
                raise ValueError(f"graph mode was captured for uint8 frames of shape {tuple(self._static_frame.shape)}, got {t.dtype} {tuple(t.shape)}")
            self._static_frame.copy_(t, non_blocking=True)
            self._graph.replay()
            pts, homo_score = self._g_pts, self._g_score
        else:
            frame = FR.upload(img)
            cx0, cy0 = self.init_pos
            if self.similarity is not None:
                dcx, dcy, scale_delta, rot_delta, best_score = self.similarity(frame, self.init_pos)
                cx, cy = cx0 + dcx, cy0 + dcy
                self.center_pos = np.array([cx, cy], np.float64)
                H_sim = torch.tensor(rot_scale_around_center_shift_tran(cx, cy, rot_delta, scale_delta, dcx, dcy),
                                     dtype=torch.float64).to(self.dev, non_blocking=True)
                params = FR._dev_f64([cx, cy, self.init_s_z_sm * scale_delta] + [float(a) for a in self.channel_average], self.dev)
                H, pts, homo_score = self._body(frame, params, H_sim, rot_delta, cx, cy)
            else:
                H, pts, homo_score = self._body(frame, self._const_params, self._eye)
            self.H_total = H
        self.last_points, self.last_score = pts, homo_score
        if not sync:
            return {"points": pts, "polygon": pts, "best_score": best_score}
        pn = pts.cpu().numpy()
        self.host_syncs += 1
        mx, mn = pn.max(0), pn.min(0)
        bbox = [mn[0], mn[1], mx[0] - mn[0], mx[1] - mn[1]]
        return {"bbox_aligned": bbox, "best_score": best_score, "polygon": pn, "points": pn, "bbox": bbox}
