"""The similarity half of the tracker's per-frame loop on the device (BASELINE configs[3]).

    TrackerConfig                      the cfg.TRACK / cfg.POINT / cfg.TRAIN values the decode uses (hdn/core/config.py)
    SimilarityDecoder                  hdn_tracker_proj_e2e.py:169-186 and :197-214 as two kernels (hdn_similarity_*_f32)
    DeviceSimilarity(model)            :157-214 around ModelBuilder.track_new / track_new_lp
                                       (hdn/models/model_builder_e2e_unconstrained_v2.py:131-158)

The reference turns the two heads' maps into (delta_cx, delta_cy, scale_delta, rot_delta) in numpy after four
.cpu().numpy() reads per frame (hdn/tracker/hdn_tracker.py:51-67,84-91, hdn/tracker/base_tracker.py:54-59).  Here the maps
stay on the device: two one-wave kernels write a float64 state record (layout: include/hdn_hip.h) holding the centre, the
gates' outcome, H_sim, the rotate-back matrix and the parameters of the next crops, and the crop / warp kernels read that record
directly.  Nothing synchronises, so the whole frame body remains capturable as one hipGraph.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np
import torch

from . import _lib
from . import frame as FR

SEQ_DOUBLES, STATE_DOUBLES = 8, 48     # HDN_SIM_SEQ_DOUBLES / HDN_SIM_STATE_DOUBLES


@dataclasses.dataclass
class TrackerConfig:
    """Defaults = hdn/core/config.py merged with experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml."""
    exemplar_size: int = 127          # cfg.TRACK.EXEMPLAR_SIZE (= cfg.TRAIN.EXEMPLAR_SIZE)
    instance_size: int = 255          # cfg.TRACK.INSTANCE_SIZE
    base_size: int = 8                # cfg.TRACK.BASE_SIZE
    stride: int = 8                   # cfg.POINT.STRIDE
    stride_lp: int = 8                # cfg.POINT.STRIDE_LP
    output_size_lp: int = 13          # cfg.TRAIN.OUTPUT_SIZE_LP
    context_amount: float = 0.5       # cfg.TRACK.CONTEXT_AMOUNT
    window_influence: float = 0.1632532824922313   # cfg.TRACK.WINDOW_INFLUENCE (config.py default: 0.45)
    cls_out_channels: int = 2         # cfg.BAN.KWARGS.cls_out_channels: 2 = softmax class 1, 1 = sigmoid (hdn_tracker.py:84-91)

    @classmethod
    def from_reference(cls, cfg):
        """From the reference's yacs node (hdn.core.config.cfg) after the experiment YAML was merged."""
        return cls(exemplar_size=int(cfg.TRACK.EXEMPLAR_SIZE), instance_size=int(cfg.TRACK.INSTANCE_SIZE), base_size=int(cfg.TRACK.BASE_SIZE),
                   stride=int(cfg.POINT.STRIDE), stride_lp=int(cfg.POINT.STRIDE_LP), output_size_lp=int(cfg.TRAIN.OUTPUT_SIZE_LP),
                   context_amount=float(cfg.TRACK.CONTEXT_AMOUNT), window_influence=float(cfg.TRACK.WINDOW_INFLUENCE),
                   cls_out_channels=int(cfg.BAN.KWARGS.cls_out_channels))

    @property
    def score_size(self) -> int:      # hdn_tracker_proj_e2e.py:24-25
        return (self.instance_size - self.exemplar_size) // self.stride + 1 + self.base_size


def generate_points(stride: int, size: int) -> np.ndarray:
    """hdnTracker.generate_points / generate_points_lp (hdn/tracker/hdn_tracker.py:32-49): float32 [size*size, 2] cell centres."""
    ori = -(size // 2) * stride
    x, y = np.meshgrid([ori + stride * dx for dx in np.arange(0, size)], [ori + stride * dy for dy in np.arange(0, size)])
    points = np.zeros((size * size, 2), dtype=np.float32)
    points[:, 0], points[:, 1] = x.astype(np.float32).flatten(), y.astype(np.float32).flatten()
    return points


def sequence_constants(init_pos, init_s_z, init_s_z_sm, channel_average, cfg: TrackerConfig) -> np.ndarray:
    """The per-sequence record the kernels read (`seq` in include/hdn_hip.h); s_x as hdn_tracker_proj_e2e.py:161,164."""
    ratio = np.round(cfg.instance_size / cfg.exemplar_size)
    s_x = np.floor(init_s_z * ratio)
    avg = [float(a) for a in np.asarray(channel_average).reshape(-1)]
    if len(avg) != 3:
        raise ValueError("the tracker's frames are 3-channel (BGR)")
    return np.array([init_pos[0], init_pos[1], init_s_z, s_x, init_s_z_sm] + avg, np.float64)


class SimilarityDecoder:
    """Device tables (Hanning window, anchor points) + the two decode launches."""

    def __init__(self, device, cfg: TrackerConfig = None):
        self.cfg = cfg or TrackerConfig()
        self.dev = torch.device(device)
        c = self.cfg
        self.S, self.S_lp = c.score_size, c.output_size_lp
        if c.cls_out_channels not in (1, 2):                          # hdn_tracker.py:84-91 has exactly these two decodes
            raise ValueError(f"cls_out_channels must be 1 (sigmoid) or 2 (softmax), got {c.cls_out_channels}")
        hanning = np.hanning(self.S)                                  # hdn_tracker_proj_e2e.py:26-29
        self.window = torch.from_numpy(np.outer(hanning, hanning).flatten()).to(self.dev)
        self.points = torch.from_numpy(generate_points(c.stride, self.S)).to(self.dev)
        self.points_lp = torch.from_numpy(generate_points(c.stride_lp, self.S_lp)).to(self.dev)
        self.mag = float(np.log(c.exemplar_size / 2) / c.exemplar_size)           # hdn_tracker.py:63
        self.rot_unit = float(np.float32(2 * np.pi / c.exemplar_size))            # :62 (float32 array * python float)

    def new_state(self, B: int = 1) -> torch.Tensor:
        return torch.zeros((B, STATE_DOUBLES), dtype=torch.float64, device=self.dev)

    @staticmethod
    def _maps(t, ch, S, name):
        if t.dim() != 4 or t.shape[1] != ch or t.shape[2] != S or t.shape[3] != S:
            raise ValueError(f"{name} must be [B,{ch},{S},{S}], got {tuple(t.shape)}")
        return t.detach().contiguous()

    def _records(self, seq, state, B):
        for t, n, name in ((seq, SEQ_DOUBLES, "seq"), (state, STATE_DOUBLES, "state")):
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or t.numel() != B * n:
                raise ValueError(f"{name} must be a contiguous float64 GPU tensor of {B} x {n}")

    def translation(self, cls, loc_c, seq, state):
        """cls [B,cls_out_channels,S,S], loc_c [B,2,S,S] -> state[:, 0:14] (include/hdn_hip.h)."""
        dev = _lib.require_device(cls, loc_c)
        B = cls.shape[0]
        cls, loc_c = self._maps(cls, self.cfg.cls_out_channels, self.S, "cls"), self._maps(loc_c, 2, self.S, "loc_c")
        self._records(seq, state, B)
        c = self.cfg
        with _lib.device_guard(dev):
            rc = _lib.load().hdn_similarity_translation_f32(_lib.ptr(cls), _lib.ptr(loc_c), _lib.ptr(self.window), _lib.ptr(self.points),
                                                            _lib.ptr(seq), _lib.ptr(state), B, self.S, c.window_influence, 8.0,
                                                            float(c.exemplar_size), c.cls_out_channels, _lib.stream_ptr(dev))   # (_convert_c hard-codes 8)
        _lib.check(rc, "similarity translation decode")

    def logpolar(self, cls_lp, loc_lp, seq, state):
        """cls_lp [B,cls_out_channels,S,S], loc_lp [B,4,S,S] + the record the translation call wrote -> state[:, 16:46]."""
        dev = _lib.require_device(cls_lp, loc_lp)
        B = cls_lp.shape[0]
        cls_lp, loc_lp = self._maps(cls_lp, self.cfg.cls_out_channels, self.S_lp, "cls_lp"), self._maps(loc_lp, 4, self.S_lp, "loc_lp")
        self._records(seq, state, B)
        with _lib.device_guard(dev):
            rc = _lib.load().hdn_similarity_logpolar_f32(_lib.ptr(cls_lp), _lib.ptr(loc_lp), _lib.ptr(self.points_lp), _lib.ptr(seq),
                                                         _lib.ptr(state), B, self.S_lp, float(self.cfg.stride_lp), self.mag, self.rot_unit,
                                                         self.cfg.cls_out_channels, _lib.stream_ptr(dev))
        _lib.check(rc, "similarity log-polar decode")


# views into one state record (1-D, length STATE_DOUBLES)
def state_fields(state_row: torch.Tensor) -> dict:
    return {"delta": state_row[0:2], "center": state_row[2:4], "stop": state_row[4], "best_score": state_row[5],
            "best_idx": state_row[6], "pscore": state_row[7], "params_moved": state_row[8:14], "scale_delta": state_row[16],
            "rot_delta": state_row[17], "best_idx_lp": state_row[18], "score_lp": state_row[19], "H_sim": state_row[20:29].view(3, 3),
            "rot_matrix": state_row[32:38], "params_homo": state_row[40:46]}


class DeviceSimilarity:
    """hdn_tracker_proj_e2e.py:157-214 for one frame, on the device.

    `model` exposes the reference's ModelBuilder interface: template(z) (:87-96), track_new(x) -> {'cls','loc_c'} (:131-140),
    track_new_lp(x, delta) -> {'cls_lp','loc_lp',...} (:144-158) — in deployment the reference's own ModelBuilder after
    hdn_amd.install.install() (PyTorch-ROCm backbone / necks, HIP correlations inside the heads, HIP log-polar sampler).

        sim = DeviceSimilarity(model, cfg)
        sim.init(frame0_u8_device, init_pos, init_s_z, init_s_z_sm, channel_average)     # template + per-sequence records
        fields = sim(stabilised_frame_u8_device)        # fills sim.state, returns views into it (no host read)
    """

    def __init__(self, model, cfg: TrackerConfig = None):
        self.model = model
        self.cfg = cfg or TrackerConfig()
        self._dec = None
        self.seq = self.state = self._params0 = None

    def decoder(self, dev) -> SimilarityDecoder:
        if self._dec is None or self._dec.dev != torch.device(dev):
            self._dec = SimilarityDecoder(dev, self.cfg)
        return self._dec

    def init(self, frame, init_pos, init_s_z, init_s_z_sm, channel_average):
        """hdnTrackerHomo.init's model side (:99-107): z_crop = get_subwindow_for_homo(img, center_pos, EXEMPLAR_SIZE, s_z, avg,
        islog=1); model.template(z_crop) — plus the records the per-frame kernels read.
        A batch of B sequences in lock step (hdn_amd.batched_tracker): frame [B,H,W,3], init_pos [B,2], init_s_z / init_s_z_sm [B],
        channel_average [B,3]; the model then holds B templates (its zf has batch B) and every per-frame call runs at batch B."""
        dev = frame.device
        dec = self.decoder(dev)
        if frame.dim() == 4:
            B = frame.shape[0]
            pos, sz, szs, avg = (np.asarray(a, np.float64) for a in (init_pos, init_s_z, init_s_z_sm, channel_average))
            if pos.shape != (B, 2) or sz.shape != (B,) or szs.shape != (B,) or avg.shape != (B, 3):
                raise ValueError(f"a batch of {B} sequences takes init_pos [{B},2], init_s_z / init_s_z_sm [{B}], channel_average [{B},3]")
            host = np.stack([sequence_constants(pos[b], sz[b], szs[b], avg[b], self.cfg) for b in range(B)])
            tparams = np.concatenate([pos, sz[:, None], avg], axis=1)          # the template crop: about init_pos, side init_s_z
        else:
            B = 1
            host = sequence_constants(init_pos, init_s_z, init_s_z_sm, channel_average, self.cfg).reshape(1, -1)
            tparams = None
        self.batch, self.batched = B, frame.dim() == 4
        self.seq = torch.from_numpy(host).to(dev)
        if not self.batched:
            self.seq = self.seq.reshape(-1)
        # get_subwindow(img, init_pos, INSTANCE_SIZE, s_x, avg) (:164-166): the first search crop never moves
        self._params0 = torch.from_numpy(np.ascontiguousarray(host[:, [0, 1, 3, 5, 6, 7]])).to(dev)
        if not self.batched:
            self._params0 = self._params0.reshape(-1)
        self.state = dec.new_state(B)
        if not self.batched:
            z_crop, _ = FR.get_subwindow_for_homo(frame, init_pos, self.cfg.exemplar_size, init_s_z, channel_average, islog=1)
        else:
            z_crop = FR.get_subwindow(frame, None, self.cfg.exemplar_size, None, None, params=torch.from_numpy(tparams).to(dev), islog=1)
        with torch.no_grad():
            self.model.template(z_crop)
        return z_crop

    def __call__(self, frame):
        if self.seq is None:
            raise RuntimeError("DeviceSimilarity.init() has not been called for this sequence")
        c, dec = self.cfg, self.decoder(frame.device)
        B, batched = getattr(self, "batch", 1), getattr(self, "batched", False)
        if (frame.shape[0] if frame.dim() == 4 else 1) != B or batched != (frame.dim() == 4):
            raise ValueError(f"init() was given {B} sequence(s); this call has {frame.shape[0] if frame.dim() == 4 else 1}")
        seq = self.seq.view(B, -1)
        # 1. translation (:164-186)
        x_crop = FR.get_subwindow(frame, None, c.instance_size, None, None, params=self._params0)
        with torch.no_grad():
            out = self.model.track_new(x_crop)
        dec.translation(out["cls"], out["loc_c"], seq, self.state)
        # 2. scale / rotation (:189-214): crop about the moved centre, log-polar head
        x_moved = FR.get_subwindow(frame, None, c.instance_size, None, None, params=self.state[:, 8:14] if batched else self.state.view(-1)[8:14])
        with torch.no_grad():
            out = self.model.track_new_lp(x_moved, [0, 0])
        dec.logpolar(out["cls_lp"], out["loc_lp"], seq, self.state)
        if batched:    # column views of the [B, 48] record (the kernels take them with the rows' stride)
            return {"rot_matrix": self.state[:, 32:38], "params_homo": self.state[:, 40:46], "H_sim": self.state[:, 20:29]}
        return state_fields(self.state.view(-1))
