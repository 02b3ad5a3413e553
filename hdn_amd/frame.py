"""Device-resident frame handling for the tracker loop.

The reference handles every frame on the host (numpy + OpenCV) and uploads three crops per frame
(hdn/tracker/base_tracker.py:134-135,211-212); here the uint8 frame is uploaded ONCE and the crops / warps are kernels:

    upload(img)                                            np.uint8 [H,W,3] (BGR) -> device tensor
    get_subwindow(frame, pos, model_sz, original_sz, avg)  <- SiameseTracker.get_subwindow, base_tracker.py:61-136
    get_subwindow_for_homo(...)                            <- base_tracker.py:138-213 (also returns the crop points)
    get_search_info(frame, pos, original_sz, avg)          <- get_subwindow_for_homo + get_search_info (get_img_info.py:42-70), fused
    warp_perspective(frame, M)                             <- cv2.warpPerspective(img, M, BORDER_REPLICATE), hdn_tracker_proj_e2e.py:154
    rot_around_center(frame, cx, cy, rot)                  <- img_rot_around_center, hdn/utils/transform.py:69-100
    get_polar_img(patch) / get_subwindow(..., islog=1)     <- getPolarImg (cv2.logPolar), hdn/models/logpolar.py:11-29, base_tracker.py:119-126

Same argument meaning and return shapes as the reference's functions; `pos` / `original_sz` / matrices may be host numbers
(packed into a small device array without synchronising) or float64 device tensors (nothing leaves the device).  The parts
that are OpenCV in the reference are restatements (parity-unpinned, see include/hdn_hip.h).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib


def upload(img) -> torch.Tensor:
    """np.uint8 [H,W,C] (as cv2.imread returns it) -> contiguous uint8 device tensor; the one host->device copy of a frame."""
    if isinstance(img, torch.Tensor):
        t = img
    else:
        a = np.ascontiguousarray(img)
        if a.dtype != np.uint8 or a.ndim != 3:
            raise TypeError(f"expected a uint8 [H,W,C] frame, got {a.dtype} {a.shape}")
        t = torch.from_numpy(a)
    if t.dtype != torch.uint8 or t.dim() != 3:
        raise TypeError("expected a uint8 [H,W,C] frame")
    if not torch.cuda.is_available():
        raise _lib.HdnHipError("hdn_amd runs on the GPU only; there is no CPU fallback")
    return t.contiguous().cuda(non_blocking=True) if not t.is_cuda else t.contiguous()


def _check_frame(frame):
    """-> (H, W, C) of a frame [H,W,C], or of every frame of a batch [B,H,W,C] (B sequences in lock step: hdn_amd.batched_tracker)."""
    if not isinstance(frame, torch.Tensor) or frame.dtype != torch.uint8 or frame.dim() not in (3, 4) or not frame.is_cuda:
        raise _lib.HdnHipError("frame must be a uint8 [H,W,C] (or [B,H,W,C]) tensor on the GPU (hdn_amd.frame.upload); there is no CPU fallback")
    if not frame.is_contiguous():
        raise ValueError("frame must be contiguous")
    return frame.shape[-3:]


def _records(t, B, width, what):
    """A float64 device array of B parameter records of `width` doubles -> (tensor to keep alive, pointer, stride in doubles).  The records
    may be a column slice of a wider per-sequence array (state[:, 8:14]): only the rows' stride has to be uniform."""
    if t.dtype != torch.float64 or not t.is_cuda:
        raise TypeError(f"{what} must be a float64 GPU tensor")
    if t.dim() == 1 and B == 1:
        t = t.reshape(1, -1)
    if t.dim() != 2 or t.shape[0] != B or t.shape[1] != width or t.stride(1) != 1 or (B > 1 and t.stride(0) < width):
        raise ValueError(f"{what} must be [{B}, {width}] with unit column stride, got {tuple(t.shape)} strides {tuple(t.stride())}")
    return t, t.data_ptr(), (t.stride(0) if B > 1 else width)


def _dev_f64(values, dev) -> torch.Tensor:
    """Host numbers -> a small float64 device array (pinned-free async copy); device tensors pass through."""
    if isinstance(values, torch.Tensor):
        if values.dtype != torch.float64 or not values.is_cuda:
            raise TypeError("device-side parameters must be float64 GPU tensors")
        return values.contiguous().reshape(-1)
    return torch.tensor([float(v) for v in values], dtype=torch.float64).to(dev, non_blocking=True)


def crop_points(pos, original_sz, im_h, im_w):
    """(context_xmin, context_ymin, context_xmax + 1, context_ymax + 1) in the padded frame, as
    get_subwindow_for_homo returns them (base_tracker.py:150-167,213): host arithmetic on host numbers."""
    sz = float(original_sz)
    c = (sz - 1) / 2
    xmin = math.floor(pos[0] - c + 0.5)
    ymin = math.floor(pos[1] - c + 0.5)
    xmax, ymax = xmin + sz - 1, ymin + sz - 1
    left, top = int(max(0., -xmin)), int(max(0., -ymin))
    return (xmin + left, ymin + top, xmax + left + 1, ymax + top + 1)


def _subwindow(frame, pos, model_sz, original_sz, avg_chans, mode, params=None):
    H, W, C = _check_frame(frame)
    dev = frame.device
    B = frame.shape[0] if frame.dim() == 4 else 1
    if params is None:
        if B != 1:
            raise ValueError("a batch of frames takes its crop parameters as a float64 device tensor [B, 3 + C] (`params`)")
        params = _dev_f64([pos[0], pos[1], original_sz] + [float(a) for a in np.asarray(avg_chans).reshape(-1)], dev)
    if B == 1 and params.numel() != 3 + C:
        raise ValueError(f"params must be [cx, cy, original_sz, avg x {C}]")
    keep, pp, stride = _records(params, B, 3 + C, "params")
    m = int(model_sz)
    out = torch.empty((B, 1 if mode else C, m, m), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_subwindow_batch_f32(_lib.ptr(frame), pp, stride, _lib.ptr(out), B, H, W, C, m, mode, _lib.stream_ptr(dev))
    _lib.check(rc, "get_subwindow")
    return out


def get_subwindow(frame, pos, model_sz, original_sz, avg_chans, params=None, islog: int = 0):
    """-> float32 [1, C, model_sz, model_sz] (uint8-valued), on the device; islog=1 appends the C log-polar channels
    (np.concatenate((im_patch, getPolarImg(im_patch)), 2), base_tracker.py:119-126) -> [1, 2C, model_sz, model_sz].
    A batch of frames [B,H,W,C] with `params` [B, 3 + C] gives [B, ...]: one launch, frame b cut with record b."""
    out = _subwindow(frame, pos, model_sz, original_sz, avg_chans, 0, params)
    if islog == 1:
        return torch.cat([out, get_polar_img(out)], dim=1)
    if islog:
        raise NotImplementedError("islog=2 (cv2.linearPolar) is not used by the shipped tracker configuration")
    return out


def log_polar_maps(w: int, h: int, center, M: float):
    """(mapx, mapy) float32 [h, w] of cv2.logPolar(src, center, M, flags) = cv::warpPolar(src, size(src), center, exp(w / M),
    flags | WARP_POLAR_LOG) as OpenCV >= 3.4.2 / 4.x builds them (imgwarp.cpp): rho along x, angle along y, all in double,
    stored as float."""
    max_radius = math.exp(w / M) if M > 0 else 1.0
    kangle = 2.0 * math.pi / h
    kmag = math.log(max_radius) / w
    rhos = (np.exp(np.arange(w, dtype=np.float64) * kmag) - 1.0).astype(np.float32).astype(np.float64)
    ang = kangle * np.arange(h, dtype=np.float64)
    mx = (rhos[None, :] * np.cos(ang)[:, None] + float(center[0])).astype(np.float32)
    my = (rhos[None, :] * np.sin(ang)[:, None] + float(center[1])).astype(np.float32)
    return mx, my


_polar_maps = {}


def get_polar_img(patch: torch.Tensor, original=None) -> torch.Tensor:
    """getPolarImg (hdn/models/logpolar.py:11-29) on a device crop [1, C, S, S] (uint8-valued float32): cv2.logPolar about
    (S // 2, S // 2) (or round(original)) with M = S / log(S / 2), INTER_LINEAR, outliers filled with 0.  Restated OpenCV."""
    if patch.dim() != 4 or patch.dtype != torch.float32:
        raise ValueError("get_polar_img takes a float32 [B, C, H, W] crop")
    dev = _lib.require_device(patch)
    Bp, C, H, W = patch.shape
    C = Bp * C          # the maps are the same for every plane: a batch of crops is just more planes
    center = (float(np.round(original[0])), float(np.round(original[1]))) if original is not None else (float(H // 2), float(W // 2))
    key = (dev, H, W, center)
    if key not in _polar_maps:
        mx, my = log_polar_maps(W, H, center, W / math.log(W / 2))
        _polar_maps[key] = (torch.from_numpy(mx).to(dev), torch.from_numpy(my).to(dev))
    mx, my = _polar_maps[key]
    src = patch.detach().contiguous()
    out = torch.empty_like(src)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_remap_linear_f32(_lib.ptr(src), _lib.ptr(mx), _lib.ptr(my), _lib.ptr(out), C, H, W, H, W, _lib.stream_ptr(dev))
    _lib.check(rc, "get_polar_img")
    return out


def get_subwindow_for_homo(frame, pos, model_sz, original_sz, avg_chans, islog: int = 0):
    H, W, _ = _check_frame(frame)
    return get_subwindow(frame, pos, model_sz, original_sz, avg_chans, islog=islog), crop_points(pos, original_sz, H, W)


def get_search_info(frame, pos, original_sz, avg_chans, model_sz: int = 127, params=None):
    """The normalised gray crop the homography head consumes, [1, 1, 127, 127] float32 on the device: get_subwindow_for_homo
    followed by get_search_info's (x - mean) / std, channel mean (float64), in one kernel."""
    return _subwindow(frame, pos, model_sz, original_sz, avg_chans, 1, params)


def _matrices(frame, M, width, what):
    B = frame.shape[0] if frame.dim() == 4 else 1
    if isinstance(M, torch.Tensor):
        m = M if (B > 1 or M.dim() == 2) else M.contiguous().reshape(1, -1)
        if B == 1 and m.numel() != width:
            raise ValueError(f"M must be {what}")
    else:
        a = np.asarray(M, np.float64)
        if a.size != B * width:
            raise ValueError(f"M must be {what}" + (f" per frame ({B} frames)" if B > 1 else ""))
        m = _dev_f64(a.reshape(-1), frame.device).reshape(B, width)
    return (B,) + _records(m, B, width, "M")


def warp_perspective(frame, M):
    """cv2.warpPerspective(frame, M, (W, H), borderMode=cv2.BORDER_REPLICATE); M: 3x3 (host array or float64 device tensor).
    A batch [B,H,W,C] takes B matrices [B, 9] (rows of a wider float64 device array are fine) and is one launch."""
    H, W, C = _check_frame(frame)
    B, keep, mp, stride = _matrices(frame, M, 9, "3x3")
    out = torch.empty_like(frame)
    with _lib.device_guard(frame.device):
        rc = _lib.load().hdn_frame_warp_perspective_batch_u8(_lib.ptr(frame), mp, stride, _lib.ptr(out), B, H, W, C, _lib.stream_ptr(frame.device))
    _lib.check(rc, "warp_perspective")
    return out


def warp_affine_cubic(frame, M):
    """cv2.warpAffine(frame, M, (W, H), flags=cv2.INTER_CUBIC, borderMode=cv2.BORDER_REPLICATE); M: 2x3 (a batch: [B, 6])."""
    H, W, C = _check_frame(frame)
    B, keep, mp, stride = _matrices(frame, M, 6, "2x3")
    out = torch.empty_like(frame)
    with _lib.device_guard(frame.device):
        rc = _lib.load().hdn_frame_warp_affine_cubic_batch_u8(_lib.ptr(frame), mp, stride, _lib.ptr(out), B, H, W, C, _lib.stream_ptr(frame.device))
    _lib.check(rc, "warp_affine_cubic")
    return out


def rot_around_center(frame, cx, cy, rot):
    """img_rot_around_center(img, cx, cy, w, h, rot) (hdn/utils/transform.py:69-100): rotation about (cx, cy), bicubic."""
    cc, ss = math.cos(rot), math.sin(rot)
    return warp_affine_cubic(frame, [cc, -ss, cx - cx * cc + cy * ss, ss, cc, cy - cy * cc - cx * ss])
