"""Drop-ins for the reference's correlation operators (hdn/core/xcorr.py), on HIP kernels.

    xcorr_depthwise(x, kernel)            <- hdn/core/xcorr.py:37-46
    xcorr_depthwise_circular(x, kernel)   <- hdn/core/xcorr.py:48-61
    xcorr_depthwise_multi(xs, kernels)    (one launch for a frame's 6 correlations)

Same names, argument order and output shapes as the reference.  Inference only: results
carry no autograd graph (the reference's inference loop never uses one).
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence

import torch

from . import _lib


def out_shape(x_shape, k_shape, circular: bool):
    """[B,C,Ho,Wo] for x[B,C,Hx,Wx] (x) k[B,C,Hk,Wk]; raises like the reference's conv2d would."""
    if len(x_shape) != 4 or len(k_shape) != 4:
        raise ValueError(f"expected 4-D [B,C,H,W] tensors, got {tuple(x_shape)} and {tuple(k_shape)}")
    B, C, Hx, Wx = x_shape
    Bk, Ck, Hk, Wk = k_shape
    if (B, C) != (Bk, Ck):
        raise ValueError(f"batch/channel mismatch: x {tuple(x_shape)} vs kernel {tuple(k_shape)}")
    if min(B, C, Hx, Wx, Hk, Wk) <= 0:
        raise ValueError(f"empty tensor: x {tuple(x_shape)}, kernel {tuple(k_shape)}")
    HP = Hx + 2 * (Hx // 2) if circular else Hx
    WP = Wx + 2 * (Wx // 2) if circular else Wx
    if Hk > HP or Wk > WP:
        raise ValueError(f"kernel {Hk}x{Wk} larger than the (padded) search plane {HP}x{WP}")
    return (B, C, HP - Hk + 1, WP - Wk + 1)


def _prep(x: torch.Tensor, kernel: torch.Tensor):
    dev = _lib.require_device(x, kernel)
    return dev, x.detach().contiguous(), kernel.detach().contiguous()


def _xcorr(x: torch.Tensor, kernel: torch.Tensor, circular: bool) -> torch.Tensor:
    shape = out_shape(x.shape, kernel.shape, circular)
    dev, xc, kc = _prep(x, kernel)
    lib = _lib.load()
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    B, C, Hx, Wx = xc.shape
    Hk, Wk = kc.shape[2:]
    fn = lib.hdn_xcorr_depthwise_circ_f32 if circular else lib.hdn_xcorr_depthwise_f32
    with torch.cuda.device(dev):
        rc = fn(_lib.ptr(xc), _lib.ptr(kc), _lib.ptr(out), B, C, Hx, Wx, Hk, Wk, _lib.stream_ptr(dev))
    _lib.check(rc, "xcorr_depthwise_circular" if circular else "xcorr_depthwise")
    return out


def xcorr_depthwise(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """depthwise cross correlation: out[b,c,i,j] = sum_uv x[b,c,i+u,j+v] * kernel[b,c,u,v]."""
    return _xcorr(x, kernel, False)


def xcorr_depthwise_circular(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """depthwise cross correlation for log-polar maps: rows wrap, columns replicate, pad = size//2."""
    return _xcorr(x, kernel, True)


def xcorr_depthwise_multi(xs: Sequence[torch.Tensor], kernels: Sequence[torch.Tensor], circular: bool = False) -> List[torch.Tensor]:
    """n same-shaped correlations in one launch (n <= 8)."""
    if len(xs) != len(kernels) or not xs:
        raise ValueError("xs and kernels must be non-empty and of equal length")
    n = len(xs)
    if n > 8:
        raise ValueError("at most 8 problems per launch")
    shape = out_shape(xs[0].shape, kernels[0].shape, circular)
    xcs, kcs = [], []
    dev = None
    for x, k in zip(xs, kernels):
        if x.shape != xs[0].shape or k.shape != kernels[0].shape:
            raise ValueError("all problems of one launch must share a shape")
        d, xc, kc = _prep(x, k)
        if dev is not None and d != dev:
            raise _lib.HdnHipError("problems on different devices")
        dev = d
        xcs.append(xc)
        kcs.append(kc)
    lib = _lib.load()
    outs = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(n)]
    arr = ctypes.c_void_p * n
    B, C, Hx, Wx = xcs[0].shape
    Hk, Wk = kcs[0].shape[2:]
    with torch.cuda.device(dev):
        rc = lib.hdn_xcorr_depthwise_multi_f32(
            arr(*[t.data_ptr() for t in xcs]), arr(*[t.data_ptr() for t in kcs]), arr(*[t.data_ptr() for t in outs]),
            n, int(bool(circular)), B, C, Hx, Wx, Hk, Wk, _lib.stream_ptr(dev),
        )
    _lib.check(rc, "xcorr_depthwise_multi")
    return outs


def last_variant() -> str:
    """Name of the kernel the last correlation call dispatched to (tests / profiles)."""
    return _lib.load().hdn_last_xcorr_variant().decode()
