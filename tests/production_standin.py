"""TEST INFRASTRUCTURE: a seeded stand-in for the reference's ModelBuilder (hdn/models/model_builder_e2e_unconstrained_v2.py:36-158)
with the PRODUCTION SHAPES of the shipped configuration (experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml):

    backbone   ResNet-50, stride 8, layers 2 / 3 / 4 used, dilation 2 / 4 in layers 3 / 4 — the layer table of
               hdn/models/backbone/resnet_atrous.py:113-199 (7x7 / s2 / p0 stem, 3x3-convolution downsample branches, un-padded
               stride-2 stage): 255 px -> 31 x 31 x (512, 1024, 2048), 127 px -> 15 x 15, 303 px -> 37 x 37
    necks      AdjustAllLayer (hdn/models/neck/neck.py:11-51): 1x1 convolution + BatchNorm to 256 channels per level; `neck` cuts
               maps narrower than 20 to their centre 7 x 7, `neck_lp` does not
    heads      MultiBAN / MultiCircBAN at 256 channels (hdn_amd.heads: the product's HIP correlations; on the CPU twin the oracle's)
    log-polar  STN_Polar (hdn_amd.logpolar on the GPU twin, the oracle's sampler on the CPU twin)
    hm_net     the homography estimator (hdn_amd.HomoModelBuilder)

and the interface hdnTrackerHomo drives: template(z) (:87-96), track_new(x) (:131-140), track_new_lp(x, delta) (:144-158),
feature_extractor, zf / zf_lp.  Written here from the layer table; the weights are seeded (no snapshot exists in the image),
BatchNorm statistics are calibrated once on crops of the synthetic sequence so that a random network keeps O(1) activations
through 50 layers.  The deployment's networks are the reference's own modules on PyTorch-ROCm: this file exists so that
hdn_amd.tracker.DeviceTrackerHomo(model) — the exact object install(tracker=True) registers — can be RUN and TIMED at the
production sizes without them.

As in tests/standin_model.py a fixed centre prior on the classification maps and small loc_scale values make an untrained model a
usable, non-trivial tracker signal.  The loc_scale values are much smaller than the toy model's: a trained tracker is a negative
feedback loop, a random 50-layer network is not — its regression outputs change by O(their size) per pixel of input shift, and
through the H_total recurrence a 1e-4 px difference between two runs of the loop grew x15 per frame at loc_scale 0.4; at 0.01 /
0.001 (0.15 px, 0.03 %, 0.1 mrad of similarity motion per frame) the loop is close to neutral (measured on the CPU twin).  With instance_size = 303 (BASELINE configs[4]) the log-polar sampler keeps its 127-point
grid (STN_Polar(255)): the reference's own STN_Polar(303) would hand its heads 18 x 18 features -> 16 x 16 maps, which its decode
(13 x 13 anchor points, hdn_tracker.py:55) cannot take.
"""
from __future__ import annotations

import copy

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------------- backbone
# Attribute names follow the reference's modules (Bottleneck: conv1 / bn1 / conv2 / bn2 / conv3 / bn3 / relu / downsample,
# resnet_atrous.py:60-108; ResNet: conv1 / bn1 / relu / maxpool / layer1..4 / used_layers, :111-199; AdjustLayer.downsample and
# AdjustAllLayer.downsample2..4, neck.py:11-51), so that whatever hdn_amd does to the reference's networks by structure
# (hdn_amd.backbone: BatchNorm folding, fused epilogues) happens to this stand-in too.
class _Unit(nn.Module):
    """1x1 reduce -> 3x3 (stride / dilation) -> 1x1 expand, BatchNorm after each, residual add, ReLU."""

    def __init__(self, cin, mid, stride=1, dil=1, pad=1, skip=None):
        super().__init__()
        cout = 4 * mid
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride=stride, padding=pad, dilation=dil, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = skip
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


def _skip(cin, cout, k, stride=1, dil=1, pad=0):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=False), nn.BatchNorm2d(cout))


def _stage(cin, mid, n, first, rest):
    """first = (stride, dil, pad, skip kernel, skip dil, skip pad) of the stage's first unit; rest = (dil, pad) of the others."""
    stride, dil, pad, sk, sdil, spad = first
    units = [_Unit(cin, mid, stride, dil, pad, _skip(cin, 4 * mid, sk, stride, sdil, spad))]
    units += [_Unit(4 * mid, mid, 1, rest[0], rest[1]) for _ in range(n - 1)]
    return nn.Sequential(*units)


class AtrousResNet50(nn.Module):
    """-> [level2 (512 ch), level3 (1024), level4 (2048)], all at stride 8."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = _stage(64, 64, 3, (1, 1, 1, 1, 1, 0), (1, 1))             # 1x1 skip
        self.layer2 = _stage(256, 128, 4, (2, 1, 0, 3, 1, 0), (1, 1))           # stride 2 without padding, 3x3 skip
        self.layer3 = _stage(512, 256, 6, (1, 1, 1, 3, 1, 1), (2, 2))           # dilation 2 (first unit: 1), 3x3 skip
        self.layer4 = _stage(1024, 512, 3, (1, 2, 2, 3, 2, 2), (4, 4))          # dilation 4 (first unit: 2), dilated 3x3 skip
        self.used_layers = [2, 3, 4]

    def forward(self, x):
        x_ = self.relu(self.bn1(self.conv1(x)))
        p1 = self.layer1(self.maxpool(x_))
        p2 = self.layer2(p1)
        p3 = self.layer3(p2)
        out = [x_, p1, p2, p3, self.layer4(p3)]
        return [out[i] for i in self.used_layers]


class _Adjust(nn.Module):
    def __init__(self, cin, cout, cut):
        super().__init__()
        self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout))
        self.cut = cut

    def forward(self, x):
        x = self.downsample(x)
        return x[:, :, 4:11, 4:11] if self.cut and x.size(3) < 20 else x


class Necks(nn.Module):
    def __init__(self, cut):
        super().__init__()
        self.num = 3
        for i, c in enumerate((512, 1024, 2048)):
            self.add_module("downsample" + str(i + 2), _Adjust(c, 256, cut))

    def forward(self, feats):
        return [getattr(self, "downsample" + str(i + 2))(f) for i, f in enumerate(feats)]


def _prior(n, cy, cx, width, amp=6.0):
    yy, xx = torch.meshgrid(torch.arange(float(n)), torch.arange(float(n)), indexing="ij")
    return amp * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / width).reshape(1, 1, n, n)


def _seed(module, seed):
    """Deterministic He-normal convolution weights; BatchNorm: identity transform, the LAST norm of every residual unit damped so
    that the residual stream does not grow by a factor per unit."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            fan = m.kernel_size[0] * m.kernel_size[1] * m.in_channels
            m.weight.data = torch.randn(m.weight.shape, generator=g) * float(np.sqrt(2.0 / fan))
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, _Unit):
            m.bn3.weight.data.fill_(0.25)


class ProductionStandIn(nn.Module):
    """The GPU twin (move it with .to(device)); `instance_size` sizes the classification prior only (25 x 25 at 255, 31 x 31 at 303)."""

    def __init__(self, hm_net, seed: int = 31, loc_scale: float = 0.01, loc_scale_lp: float = 0.001, instance_size: int = 255,
                 cls_gain: float = 0.25, cls_gain_lp: float = 0.05):
        super().__init__()
        from hdn_amd import heads as HD
        from hdn_amd.logpolar import STN_Polar
        from standin_model import _seed_head
        torch.manual_seed(seed)
        self.backbone = AtrousResNet50()
        self.neck, self.neck_lp = Necks(cut=True), Necks(cut=False)
        _seed(self.backbone, seed + 1), _seed(self.neck, seed + 2), _seed(self.neck_lp, seed + 3)
        self.head = _seed_head(HD.MultiBAN([256] * 3, 2, weighted=True), seed + 4, loc_scale)
        self.head_lp = _seed_head(HD.MultiCircBAN([256] * 3, 2, weighted=True), seed + 5, loc_scale_lp)
        for head, gain in ((self.head, cls_gain), (self.head_lp, cls_gain_lp)):     # random classification maps well below the prior
            for box in (head.box2, head.box3, head.box4):
                box.cls.head[3].weight.data.mul_(gain), box.cls.head[3].bias.data.mul_(gain)
        self.logpolar_instance = STN_Polar(255)
        self.hm_net = hm_net
        S = (instance_size - 127) // 8 + 1 + 8
        self.register_buffer("cls_prior", _prior(S, S // 2 + 0.6, S // 2 - 0.3, 18.0))
        self.register_buffer("cls_prior_lp", _prior(13, 6.2, 5.9, 8.0))
        self.zf = self.zf_lp = None
        self._polar0 = None

    # -- calibration: BatchNorm statistics of backbone and necks from a few crops (once, on the CPU, before .to(device))
    @torch.no_grad()
    def calibrate(self, crops_255, crops_127):
        """crops_*: float32 [n,3,S,S], 0..255 valued (what get_subwindow returns)."""
        mods = [m for part in (self.backbone, self.neck, self.neck_lp) for m in part.modules() if isinstance(m, nn.BatchNorm2d)]
        for m in mods:
            m.momentum, m.training = 1.0, True          # running statistics := this batch's statistics
        for x in (crops_127, crops_255):                # (the search-size statistics are the ones that stay)
            f = self.backbone(x)
            self.neck(f), self.neck_lp(f)
        for m in mods:
            m.momentum, m.training = 0.1, False
        return self.eval()

    def feature_extractor(self, x):
        return self.backbone(x)

    def template(self, z):
        self.zf = [f.contiguous() for f in self.neck(self.feature_extractor(z[:, 0:3]))]
        self.zf_lp = [f.contiguous() for f in self.neck_lp(self.feature_extractor(z[:, 3:6]))]

    def track_new(self, x, delta=[0, 0]):
        cls, loc_c = self.head(self.zf, self.neck(self.feature_extractor(x)))
        return {"cls": torch.cat([cls[:, 0:1], cls[:, 1:2] + self.cls_prior], dim=1), "loc_c": loc_c}

    def track_new_lp(self, x, delta=[0, 0]):
        polar = self._polar0
        if polar is None or polar.device != x.device or polar.shape[0] != x.shape[0]:
            polar = self._polar0 = torch.zeros((x.shape[0], 2), dtype=torch.float32, device=x.device)
        x_lp, grid = self.logpolar_instance(x, polar, delta)
        cls_lp, loc_lp = self.head_lp(self.zf_lp, self.neck_lp(self.feature_extractor(x_lp)))
        return {"x_lp": x_lp, "cls_lp": torch.cat([cls_lp[:, 0:1], cls_lp[:, 1:2] + self.cls_prior_lp], dim=1), "loc_lp": loc_lp, "grid": grid}


class ProductionStandInCPU:
    """Same weights on the CPU: PyTorch-CPU convolutions for backbone and necks, the oracle's heads and log-polar sampler
    (pinned to the reference's goldens)."""

    def __init__(self, twin: ProductionStandIn):
        from oracle import hdn_oracle as O
        self.O = O
        cpu = lambda m: copy.deepcopy(m).cpu().eval()
        self.backbone, self.neck, self.neck_lp = cpu(twin.backbone), cpu(twin.neck), cpu(twin.neck_lp)
        self.sd = {k: v.detach().cpu().clone() for k, v in twin.head.state_dict().items()}
        self.sd_lp = {k: v.detach().cpu().clone() for k, v in twin.head_lp.state_dict().items()}
        self.cls_prior, self.cls_prior_lp = twin.cls_prior.detach().cpu().clone(), twin.cls_prior_lp.detach().cpu().clone()

    def template(self, z):
        self.zf = self.neck(self.backbone(z[:, 0:3]))
        self.zf_lp = self.neck_lp(self.backbone(z[:, 3:6]))

    def track_new(self, x, delta=[0, 0]):
        cls, loc_c = self.O.multi_ban(self.zf, self.neck(self.backbone(x)), self.sd, circular=False)
        return {"cls": torch.cat([cls[:, 0:1], cls[:, 1:2] + self.cls_prior], dim=1), "loc_c": loc_c}

    def track_new_lp(self, x, delta=[0, 0]):
        x_lp, grid = self.O.logpolar_sample(x, torch.zeros((x.shape[0], 2)), delta, image_sz=255)
        cls_lp, loc_lp = self.O.multi_ban(self.zf_lp, self.neck_lp(self.backbone(x_lp)), self.sd_lp, circular=True)
        return {"x_lp": x_lp, "cls_lp": torch.cat([cls_lp[:, 0:1], cls_lp[:, 1:2] + self.cls_prior_lp], dim=1), "loc_lp": loc_lp, "grid": grid}


def calibration_crops(frames, init, n=3):
    """A few 255-px / 127-px crops about the target of the first frames (oracle crop code: pinned to the reference's)."""
    from oracle import frame_oracle as FO
    poly = init["poly"]
    s_z = float(np.floor(np.sqrt((poly[2] + 0.5 * (poly[2] + poly[3])) * (poly[3] + 0.5 * (poly[2] + poly[3])))))
    c255, c127 = [], []
    for f in frames[:n]:
        avg = np.mean(f, axis=(0, 1))
        c255.append(FO.get_subwindow(f, np.array(poly[:2]), 255, np.floor(2 * s_z), avg)[0])
        c127.append(FO.get_subwindow(f, np.array(poly[:2]), 127, s_z, avg)[0])
    return torch.from_numpy(np.stack(c255)), torch.from_numpy(np.stack(c127))
