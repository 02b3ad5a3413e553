"""STN_Polar on a HIP kernel.

    STN_Polar(image_sz).forward(x, polar, delta=[0, 0]) -> (x_lp, grid)     <- hdn/models/logpolar.py:50-134

Same constructor / forward signature and return values as the reference module.  The reference rebuilds the
sampling grid on the CPU and uploads it on every call (logpolar.py:121,110-111); here the 1-D factors of the grid
are cached on the device per rotation offset and the grid point is formed inside the sampling kernel.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib


def tables(size: int, rot: float = 0.0):
    """rho[r], cos(theta[a]), sin(theta[a]) exactly as STN_Polar._prepare_grid computes them (CPU, fp32)."""
    ls = torch.linspace(0, size - 1, size)
    mag = math.log(size / 2) / size
    rho = torch.exp(mag * ls) - 1.0
    theta = ls * 2.0 * math.pi / size + rot
    return rho, torch.cos(theta), torch.sin(theta)


def logpolar_sample(x: torch.Tensor, polar: torch.Tensor, tabs, want_grid: bool = True):
    """x [B,C,H,W], polar [B,2], tabs = device tensors (rho, cos, sin) of length S -> (x_lp [B,C,S,S], grid or None)."""
    if x.dim() != 4:
        raise ValueError(f"x must be [B,C,H,W], got {tuple(x.shape)}")
    B, C, H, W = x.shape
    rho, c, s = tabs
    S = rho.numel()
    if tuple(polar.shape) == (1, 2) and B > 1:
        # the reference broadcasts one origin over the batch (`x.repeat([batch, 1, 1]) + polar[:, 0]...`, hdn/models/logpolar.py:113-114;
        # ModelBuilder.track_new_lp always passes a [1, 2] zero origin, model_builder_e2e_unconstrained_v2.py:145)
        polar = polar.expand(B, 2)
    if tuple(polar.shape) != (B, 2):
        raise ValueError(f"polar must be [{B},2] (or [1,2], broadcast), got {tuple(polar.shape)}")
    dev = _lib.require_device(x, polar, rho, c, s)
    xc, pc = x.detach().contiguous(), polar.detach().contiguous()
    out = torch.empty((B, C, S, S), dtype=torch.float32, device=dev)
    grid = torch.empty((B, S, S, 2), dtype=torch.float32, device=dev) if want_grid else None
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_logpolar_sample_f32(
            _lib.ptr(xc), _lib.ptr(pc), _lib.ptr(rho), _lib.ptr(c), _lib.ptr(s), _lib.ptr(out),
            _lib.ptr(grid) if want_grid else None, B, C, H, W, S, _lib.stream_ptr(dev))
    _lib.check(rc, "STN_Polar")
    return out, grid


class STN_Polar(nn.Module):
    """Drop-in for hdn.models.logpolar.STN_Polar (inference)."""

    def __init__(self, image_sz):
        super().__init__()
        self._orignal_sz = [image_sz // 2, image_sz // 2]  # (sic) the reference's attribute name
        self._tabs = {}

    def _tables(self, device, rot: float):
        key = (str(device), float(rot))
        if key not in self._tabs:
            if len(self._tabs) > 64:  # update_template() passes arbitrary rotations: keep the cache bounded
                self._tabs.clear()
            self._tabs[key] = tuple(t.to(device) for t in tables(self._orignal_sz[0], float(rot)))
        return self._tabs[key]

    def forward(self, x, polar, delta=[0, 0]):
        # like the reference, the S x S grid (S = image_sz // 2) is sampled from whatever H x W crop arrives
        # (ModelBuilder.update_template passes the 127 x 127 template through STN_Polar(255), model_builder…:98-107)
        if x.dim() != 4 or x.shape[-1] < 2 or x.shape[-2] < 2:
            raise ValueError(f"STN_Polar needs a [B,C,H,W] crop with H, W >= 2, got {tuple(x.shape)}")
        rot = float(delta[1])
        return logpolar_sample(x, polar, self._tables(x.device, rot))
