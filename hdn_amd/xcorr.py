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
    with _lib.device_guard(dev):
        rc = fn(_lib.ptr(xc), _lib.ptr(kc), _lib.ptr(out), B, C, Hx, Wx, Hk, Wk, _lib.stream_ptr(dev))
    _lib.check(rc, "xcorr_depthwise_circular" if circular else "xcorr_depthwise")
    return out


def xcorr_depthwise(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """depthwise cross correlation: out[b,c,i,j] = sum_uv x[b,c,i+u,j+v] * kernel[b,c,u,v]."""
    return _xcorr(x, kernel, False)


def xcorr_depthwise_circular(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """depthwise cross correlation for log-polar maps: rows wrap, columns replicate, pad = size//2."""
    return _xcorr(x, kernel, True)


def xcorr_depthwise_multi(xs: Sequence[torch.Tensor], kernels: Sequence[torch.Tensor], circular: bool = False,
                          outs: Sequence[torch.Tensor] = None) -> List[torch.Tensor]:
    """n same-shaped correlations in one launch (n <= 8).  `outs`: n preallocated contiguous float32 result tensors (e.g. the
    slices of one stacked buffer) instead of fresh ones."""
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
    if outs is None:
        outs = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(n)]
    else:
        outs = list(outs)
        if len(outs) != n or any(tuple(o.shape) != tuple(shape) or o.dtype != torch.float32 or o.device != dev or not o.is_contiguous()
                                 for o in outs):
            raise ValueError(f"outs must be {n} contiguous float32 tensors of shape {tuple(shape)} on {dev}")
    arr = ctypes.c_void_p * n
    B, C, Hx, Wx = xcs[0].shape
    Hk, Wk = kcs[0].shape[2:]
    with _lib.device_guard(dev):
        rc = lib.hdn_xcorr_depthwise_multi_f32(
            arr(*[t.data_ptr() for t in xcs]), arr(*[t.data_ptr() for t in kcs]), arr(*[t.data_ptr() for t in outs]),
            n, int(bool(circular)), B, C, Hx, Wx, Hk, Wk, _lib.stream_ptr(dev),
        )
    _lib.check(rc, "xcorr_depthwise_multi")
    return outs


def xcorr_fast(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """group conv2d to calculate cross correlation (hdn/core/xcorr.py:26-34): x [B,C,H,W], kernel [B,O*C,h,w] ->
    [B,O,H-h+1,W-w+1], contracting the channels.  Only the unselected UPChannelBAN head uses it (O = 2 or 4)."""
    if x.dim() != 4 or kernel.dim() != 4 or x.shape[0] != kernel.shape[0]:
        raise ValueError(f"expected x [B,C,H,W] and kernel [B,O*C,h,w], got {tuple(x.shape)} and {tuple(kernel.shape)}")
    B, C, Hx, Wx = x.shape
    OC, Hk, Wk = kernel.shape[1:]
    if C <= 0 or OC % C != 0 or OC == 0:
        raise ValueError(f"kernel channels {OC} are not a multiple of the search channels {C}")
    O = OC // C
    if O > 8:
        raise ValueError("at most 8 output channels")
    if Hk > Hx or Wk > Wx or min(B, Hx, Wx, Hk, Wk) <= 0:
        raise ValueError(f"kernel {Hk}x{Wk} does not fit the search plane {Hx}x{Wx}")
    dev, xc, kc = _prep(x, kernel)
    out = torch.empty((B, O, Hx - Hk + 1, Wx - Wk + 1), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_xcorr_fast_f32(_lib.ptr(xc), _lib.ptr(kc), _lib.ptr(out), B, C, O, Hx, Wx, Hk, Wk,
                                            _lib.stream_ptr(dev))
    _lib.check(rc, "xcorr_fast")
    return out


def xcorr_slow(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """for-loop cross correlation (hdn/core/xcorr.py:10-23): per batch element conv2d(x[i], kernel[i]) with ALL kernel
    channels contracted, i.e. kernel must be [B,C,h,w] and the result is [B,1,Ho,Wo] (= xcorr_fast with O = 1)."""
    if x.dim() == 4 and kernel.dim() == 4 and kernel.shape[1] != x.shape[1]:
        raise ValueError(f"xcorr_slow contracts every kernel channel: kernel {tuple(kernel.shape)} vs x {tuple(x.shape)}")
    return xcorr_fast(x, kernel)


def last_variant() -> str:
    """Name of the kernel the last correlation call dispatched to (tests / profiles)."""
    return _lib.load().hdn_last_xcorr_variant().decode()


NORTH_VARIANTS = {"fft": 5, "fftc": 5, "direct": 1}  # HDN_NORTH_* in include/hdn_hip.h


class north_variant:
    """Select the kernel used for the 31x31 (x) 61x61 shape, process-wide; usable as a context manager.

        with north_variant("direct"): y = xcorr_depthwise(x, k)
    """

    def __init__(self, name: str):
        if name not in NORTH_VARIANTS:
            raise ValueError(f"north variant must be one of {sorted(NORTH_VARIANTS)}")
        rc = _lib.load().hdn_xcorr_north_variant(NORTH_VARIANTS[name])
        if rc < 0:
            _lib.check(rc, "hdn_xcorr_north_variant")
        self.prev = rc

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        _lib.load().hdn_xcorr_north_variant(self.prev)
        return False


def current_north_variant() -> str:
    v = _lib.load().hdn_xcorr_north_variant(-1)
    return next(k for k, n in NORTH_VARIANTS.items() if n == v and k != "fftc")
