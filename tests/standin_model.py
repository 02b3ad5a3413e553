"""TEST INFRASTRUCTURE: a small seeded stand-in for the reference's ModelBuilder (hdn/models/model_builder_e2e_unconstrained_v2.py)
with the same interface the tracker uses — template(z) (:87-96), track_new(x) (:131-140), track_new_lp(x, delta) (:144-158),
zf / zf_lp, hm_net — so that the device-resident tracker loop can be exercised end to end without the reference's ResNet-50
backbone (which is PyTorch-ROCm's job in deployment and is not shipped here).

    StandInSiamese(hm_net)              the GPU twin: hdn_amd.heads (HIP correlations) + hdn_amd.STN_Polar (HIP sampler)
    StandInSiameseCPU(twin)             the CPU twin: the oracle's multi_ban / logpolar_sample on the same weights

Backbone: three conv levels (15x15, stride 8: 255 -> 31, 127 -> 15, like the reference's stride-8 ResNet-50 levels); neck: the
centre crop of AdjustAllLayer for templates (15 -> 7), none for the log-polar branch.  So that a seeded, untrained model makes a
usable tracker signal, the classification map gets a fixed centre prior and loc_scale is small: the decoded motion stays within
a few pixels / per cent / hundredths of a radian per frame, non-trivial in every component."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

C = 16


def _levels(seed):
    g = torch.Generator().manual_seed(seed)
    convs = nn.ModuleList([nn.Conv2d(3, C, 15, stride=8) for _ in range(3)])
    for cv in convs:
        cv.weight.data = torch.randn(cv.weight.shape, generator=g) * (0.05 / 255.0)
        cv.bias.data = torch.randn(cv.bias.shape, generator=g) * 0.1
    return convs


def _seed_head(head, seed, loc_scale):
    g = np.random.default_rng(seed)
    for m in head.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.data = torch.from_numpy(g.uniform(-0.2, 0.2, m.num_features).astype(np.float32))
            m.running_var.data = torch.from_numpy(g.uniform(0.8, 1.2, m.num_features).astype(np.float32))
    head.loc_scale.data = torch.full((3,), float(loc_scale))
    return head


class StandInSiamese(nn.Module):
    def __init__(self, hm_net, seed: int = 11, loc_scale: float = 0.4, loc_scale_lp: float = 0.05, cls_out: int = 2):
        super().__init__()
        from hdn_amd import heads as HD
        from hdn_amd.logpolar import STN_Polar
        torch.manual_seed(seed)
        self.backbone = _levels(seed + 1)
        self.head = _seed_head(HD.MultiBAN([C] * 3, cls_out, weighted=True), seed + 2, loc_scale)
        self.head_lp = _seed_head(HD.MultiCircBAN([C] * 3, cls_out, weighted=True), seed + 3, loc_scale_lp)
        self.logpolar_instance = STN_Polar(255)
        self.hm_net = hm_net
        yy, xx = torch.meshgrid(torch.arange(25.0), torch.arange(25.0), indexing="ij")
        self.register_buffer("cls_prior", 6.0 * torch.exp(-((yy - 12.6) ** 2 + (xx - 11.7) ** 2) / 18.0).reshape(1, 1, 25, 25))
        yy, xx = torch.meshgrid(torch.arange(13.0), torch.arange(13.0), indexing="ij")
        self.register_buffer("cls_prior_lp", 6.0 * torch.exp(-((yy - 6.2) ** 2 + (xx - 5.9) ** 2) / 8.0).reshape(1, 1, 13, 13))
        self.zf = self.zf_lp = None
        self.cls_out = cls_out     # cfg.BAN.KWARGS.cls_out_channels: 2 (shipped) or 1 (the sigmoid decode, hdn_tracker.py:85-87)

    def feature_extractor(self, x):
        return [F.relu(cv(x)) for cv in self.backbone]

    @staticmethod
    def with_prior(cls, prior):
        """The centre prior on the class-1 logits (2 channels) / on the single logit map (1 channel; -3: sigmoid stays low off the peak)."""
        if cls.shape[1] == 1:
            return cls + prior - 3.0
        return torch.cat([cls[:, 0:1], cls[:, 1:2] + prior], dim=1)

    @staticmethod
    def neck(feats):   # AdjustLayer's centre crop for maps smaller than 20 (hdn/models/neck/neck.py): 15 -> 7
        return [f[:, :, 4:11, 4:11] if f.shape[3] < 20 else f for f in feats]

    def template(self, z):
        self.zf = [f.contiguous() for f in self.neck(self.feature_extractor(z[:, 0:3]))]
        self.zf_lp = self.feature_extractor(z[:, 3:6])

    def track_new(self, x, delta=[0, 0]):
        cls, loc_c = self.head(self.zf, self.neck(self.feature_extractor(x)))
        return {"cls": self.with_prior(cls, self.cls_prior), "loc_c": loc_c}

    def track_new_lp(self, x, delta=[0, 0]):
        polar = getattr(self, "_polar0", None)
        if polar is None or polar.device != x.device or polar.shape[0] != x.shape[0]:
            polar = torch.zeros((x.shape[0], 2), dtype=torch.float32, device=x.device)
            self._polar0 = polar
        x_lp, grid = self.logpolar_instance(x, polar, delta)
        cls_lp, loc_lp = self.head_lp(self.zf_lp, self.feature_extractor(x_lp))
        return {"x_lp": x_lp, "cls_lp": self.with_prior(cls_lp, self.cls_prior_lp), "loc_lp": loc_lp, "grid": grid}


class StandInSiameseCPU:
    """Same weights, the oracle's ops (PyTorch-CPU restatements pinned to the reference's goldens)."""

    def __init__(self, twin: StandInSiamese):
        from oracle import hdn_oracle as O
        self.O = O
        self.convs = [(cv.weight.detach().cpu().clone(), cv.bias.detach().cpu().clone()) for cv in twin.backbone]
        self.sd = {k: v.detach().cpu().clone() for k, v in twin.head.state_dict().items()}
        self.sd_lp = {k: v.detach().cpu().clone() for k, v in twin.head_lp.state_dict().items()}
        self.cls_prior, self.cls_prior_lp = twin.cls_prior.detach().cpu().clone(), twin.cls_prior_lp.detach().cpu().clone()

    def feature_extractor(self, x):
        return [F.relu(F.conv2d(x, w, b, stride=8)) for w, b in self.convs]

    def template(self, z):
        self.zf = StandInSiamese.neck(self.feature_extractor(z[:, 0:3]))
        self.zf_lp = self.feature_extractor(z[:, 3:6])

    def track_new(self, x, delta=[0, 0]):
        cls, loc_c = self.O.multi_ban(self.zf, StandInSiamese.neck(self.feature_extractor(x)), self.sd, circular=False)
        return {"cls": StandInSiamese.with_prior(cls, self.cls_prior), "loc_c": loc_c}

    def track_new_lp(self, x, delta=[0, 0]):
        x_lp, grid = self.O.logpolar_sample(x, torch.zeros((x.shape[0], 2)), delta, image_sz=255)
        cls_lp, loc_lp = self.O.multi_ban(self.zf_lp, self.feature_extractor(x_lp), self.sd_lp, circular=True)
        return {"x_lp": x_lp, "cls_lp": StandInSiamese.with_prior(cls_lp, self.cls_prior_lp), "loc_lp": loc_lp, "grid": grid}
