"""PreShareFeature on a fused HIP kernel.

Reference: homo_estimator/Deep_homography/Oneline_DLTv1/preprocess/input_feature_extractor.py:3-29.
The module keeps the reference's parameter names (ShareFeature.{0,1,3,4,6,7}.*) so snapshots load
unchanged (hdn/utils/model_load.py:78, strict=False).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib

BN_EPS = 1e-5
N_PARAMS = 422  # HDN_SF_PARAMS in include/hdn_hip.h
_SLOTS = ((0, 1), (3, 4), (6, 7))  # (conv, bn) positions inside the nn.Sequential


def fold_params(state_dict: dict, prefix: str = "ShareFeature.") -> torch.Tensor:
    """Host-side folding of the 3 conv weights + eval-mode BatchNorms into the kernel's parameter block.

    Layout: include/hdn_hip.h (w1t[k][co], w2p[ci/2][k][co][ci%2], w3p[ci/2][k][ci%2], alpha[13], beta[13]).
    BatchNorm folding follows PyTorch's CPU eval path bit for bit:
        invstd = 1/sqrt(var + eps)        (fp32)
        alpha  = gamma * invstd           (fp32)
        beta   = bias - mean * alpha      (one rounding)
    Returns a CPU float32 tensor of N_PARAMS values.
    """
    g = lambda slot, name: state_dict[f"{prefix}{slot}.{name}"].detach().to("cpu", torch.float32)
    w1, w2, w3 = g(0, "weight"), g(3, "weight"), g(6, "weight")
    if tuple(w1.shape) != (4, 1, 3, 3) or tuple(w2.shape) != (8, 4, 3, 3) or tuple(w3.shape) != (1, 8, 3, 3):
        raise ValueError("unexpected PreShareFeature conv shapes: %s %s %s" % (tuple(w1.shape), tuple(w2.shape), tuple(w3.shape)))
    w1t = w1.reshape(4, 9).t().contiguous()                     # [k][co]
    w2t = w2.reshape(8, 2, 2, 9).permute(1, 3, 0, 2).contiguous()   # [ci/2][k][co][ci%2]
    w3t = w3.reshape(4, 2, 9).permute(0, 2, 1).contiguous()         # [ci/2][k][ci%2]
    alphas, betas = [], []
    for _, bn in _SLOTS:
        invstd = 1.0 / torch.sqrt(g(bn, "running_var") + BN_EPS)
        alpha = g(bn, "weight") * invstd
        beta = (g(bn, "bias").double() - g(bn, "running_mean").double() * alpha.double()).float()
        alphas.append(alpha)
        betas.append(beta)
    out = torch.cat([w1t.reshape(-1), w2t.reshape(-1), w3t.reshape(-1)] + alphas + betas)
    assert out.numel() == N_PARAMS
    return out


def share_feature(x: torch.Tensor, folded: torch.Tensor) -> torch.Tensor:
    """x [B,1,H,W] -> [B,1,H,W] through the fused kernel; `folded` is fold_params(...) on x's device."""
    if x.dim() != 4 or x.shape[1] != 1:
        raise ValueError(f"PreShareFeature expects [B,1,H,W], got {tuple(x.shape)}")
    dev = _lib.require_device(x, folded)
    if folded.numel() != N_PARAMS:
        raise ValueError(f"folded parameter block must hold {N_PARAMS} floats")
    xc = x.detach().contiguous()
    B, _, H, W = xc.shape
    if B == 0:
        raise ValueError("empty batch")
    out = torch.empty_like(xc)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_share_feature_f32(_lib.ptr(xc), _lib.ptr(folded), _lib.ptr(out), B, H, W, _lib.stream_ptr(dev))
    _lib.check(rc, "PreShareFeature")
    return out


class PreShareFeature(nn.Module):
    """Same structure and parameter names as the reference module; eval-mode forward runs the HIP kernel.

    In training mode (batch statistics, autograd) the stock nn.Sequential runs on the device through
    PyTorch-ROCm; training is outside this build's scope and is kept only so the module stays usable there.
    """

    def __init__(self):
        super().__init__()
        self.ShareFeature = nn.Sequential(
            nn.Conv2d(1, 4, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(4), nn.ReLU(inplace=True),
            nn.Conv2d(4, 8, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(8), nn.ReLU(inplace=True),
            nn.Conv2d(8, 1, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(1), nn.ReLU(inplace=True),
        )
        for m in self.modules():  # input_feature_extractor.py:20-25
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self._folded = None
        self._folded_key = None

    def _param_key(self, device):
        # (references to the owning modules' parameter / buffer dictionaries: a state_dict() walk per call is ~15 us of Python)
        refs = self.__dict__.get("_key_refs")
        if refs is None:
            refs = []
            for sub in self.ShareFeature.modules():
                refs += [(sub._parameters, n) for n, t in sub._parameters.items() if t is not None]
                refs += [(sub._buffers, n) for n, t in sub._buffers.items() if t is not None]
            self.__dict__["_key_refs"] = refs
        return (str(device),) + tuple((id(t), t.data_ptr(), t._version) for t in (d[n] for d, n in refs))

    def folded(self, device) -> torch.Tensor:
        """Folded parameter block on `device`, rebuilt when any parameter / buffer changed."""
        key = self._param_key(device)
        if self._folded is None or key != self._folded_key:
            self._folded = fold_params(self.ShareFeature.state_dict(), prefix="").to(device)
            self._folded_key = key
        return self._folded

    def refresh(self):
        """Drop the cached folded block.  Needed only after writing parameters through `.data`, which bypasses
        the tensor version counters the cache is keyed on (load_state_dict / .to() / in-place ops are tracked)."""
        self._folded = None
        self._folded_key = None

    def train(self, mode: bool = True):
        self.refresh()
        return super().train(mode)

    def forward(self, x):
        if self.training:
            return self.ShareFeature(x)
        return share_feature(x, self.folded(x.device))
