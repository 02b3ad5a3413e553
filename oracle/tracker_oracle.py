"""TEST INFRASTRUCTURE ONLY: CPU restatement (numpy + PyTorch-CPU) of the tracker loop hdn_amd.tracker.HomoTracker runs on
the device, i.e. of hdnTrackerHomo.init / track_new (hdn/tracker/hdn_tracker_proj_e2e.py:60-120,141-285) for the stages in
scope, with the similarity estimate fixed to the identity.  Built from oracle/frame_oracle.py (crops pinned, OpenCV pieces
restated and unpinned) and oracle/hdn_oracle.py (head pinned to the reference's goldens)."""
from __future__ import annotations

import numpy as np
import torch

from . import frame_oracle as F
from . import hdn_oracle as O


class HomoTrackerOracle:
    def __init__(self, sf_sd: dict, regress, iterations: int = 1, score_gate: float = 2.5):
        self.sf_sd, self.regress, self.iterations, self.score_gate = sf_sd, regress, iterations, score_gate

    def init(self, img, bbox, poly, gt_points, first_point=None):
        self.init_pos = np.array([poly[0], poly[1]], np.float64)
        self.size = np.array([poly[2], poly[3]], np.float64)
        self.init_s_z_sm = float(np.floor(np.sqrt(self.size[0] * self.size[1])))
        self.channel_average = np.mean(img, axis=(0, 1))
        crop, self.z_crop_points_sm = F.get_subwindow_for_homo(img, self.init_pos, 127, self.init_s_z_sm, self.channel_average)
        self.init_homo_tmp = F.search_info(crop[0])                      # float64 [1,127,127]
        self.init_points = np.asarray(gt_points, np.float32).reshape(-1, 2)
        self.H_total = np.eye(3, dtype=np.float32)

    def track_new(self, fr_idx, img):
        if np.linalg.det(self.H_total) == 0:
            self.H_total = np.eye(3, dtype=np.float32)
        img = F.warp_perspective_u8(img, np.linalg.inv(self.H_total))
        cx, cy = self.init_pos
        H_sim = np.eye(3)
        # rot_delta = 0: img_rot_around_center is the bicubic identity
        crop, _ = F.get_subwindow_for_homo(img, self.init_pos, 127, self.init_s_z_sm * 1.0, self.channel_average)
        search = F.search_info(crop[0])
        tmpl = torch.from_numpy(self.init_homo_tmp).float().unsqueeze(0)
        srch = torch.from_numpy(search).float().unsqueeze(0)
        with torch.no_grad():
            H_comp, score, _, _ = O.homo_refine(tmpl, srch, self.sf_sd, self.regress, self.iterations)
        H_hm_comp = H_comp[0]
        cw = self.z_crop_points_sm[2] - self.z_crop_points_sm[0] + 1
        ch = self.z_crop_points_sm[3] - self.z_crop_points_sm[1] + 1
        S = np.array([[127 / cw, 0, 0], [0, 127 / ch, 0], [0, 0, 1]]).astype(np.float32)
        H_hm_comp = np.linalg.inv(S) @ H_hm_comp @ S
        Sh = np.array([[1, 0, -self.z_crop_points_sm[0]], [0, 1, -self.z_crop_points_sm[1]], [0, 0, 1]]).astype(np.float32)
        H_homo = np.linalg.inv(Sh) @ H_hm_comp @ Sh
        H = self.H_total @ H_sim if float(score) > self.score_gate else self.H_total @ H_sim @ H_homo
        H = (1.0 / H.item(8)) * H
        self.H_total = H
        p = np.concatenate([self.init_points.astype(np.float64), np.ones((len(self.init_points), 1))], 1) @ H.T
        pts = (p[:, :2] / p[:, 2:3]).astype(np.float32)
        return {"points": pts, "polygon": pts, "score": float(score)}
