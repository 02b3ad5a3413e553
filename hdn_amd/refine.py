"""The tracker's homography refinement loop on the device (BASELINE config 5: large displacements, 2 iterations).

    homo_refine(net, template, search, iterations=2)   <- hdn/tracker/hdn_tracker_proj_e2e.py:242-250
    refine_warp(H_mat, search, H_comp)                 <- one step: inv / normalise H, cv2.warpPerspective(search, inv(H_hm),
                                                          borderMode=BORDER_REPLICATE), H_comp @= H_hm

The reference runs this loop on the host (numpy + OpenCV) with two device->host syncs per iteration (:244-245) and a
trip count hard-coded to 1; here every step stays on the device and nothing synchronises.  The OpenCV sampler is
restated, not linked (no cv2 in this image): see include/hdn_hip.h, hdn_refine_warp_f32 — parity-unpinned.
"""
from __future__ import annotations

import torch

from . import _lib
from .homo_model import track_proj, track_proj_pair


def refine_warp(H_mat: torch.Tensor, search: torch.Tensor, H_comp: torch.Tensor = None):
    """H_mat [B,3,3] fp32, search [B,1,H,W] fp32, H_comp [B,3,3] fp64 (updated IN PLACE) -> warped [B,1,H,W]."""
    if search.dim() != 4 or search.shape[1] != 1:
        raise ValueError(f"search must be [B,1,H,W], got {tuple(search.shape)}")
    B, _, H, W = search.shape
    if H_mat.numel() != B * 9:
        raise ValueError(f"H_mat must hold {B} 3x3 matrices, got {tuple(H_mat.shape)}")
    dev = _lib.require_device(H_mat, search)
    if H_comp is not None:
        if H_comp.dtype != torch.float64 or H_comp.numel() != B * 9 or not H_comp.is_contiguous() or H_comp.device != dev:
            raise ValueError("H_comp must be a contiguous float64 [B,3,3] tensor on the same device")
    hm, sc = H_mat.detach().reshape(B, 9).contiguous(), search.detach().contiguous()
    out = torch.empty_like(sc)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_refine_warp_f32(_lib.ptr(hm), _lib.ptr(sc), _lib.ptr(out),
                                             _lib.ptr(H_comp) if H_comp is not None else None, B, H, W, _lib.stream_ptr(dev))
    _lib.check(rc, "refine_warp")
    return out


_CONST = {}


def _constants(dev, B, H, W):
    """h4p / patch_indices of the full-patch case (get_img_info.py:88-98), built once per (device, batch, size): a
    torch.tensor(list, device=...) is a host->device copy, which a hipGraph capture cannot contain."""
    key = (str(dev), B, H, W)
    if key not in _CONST:
        if len(_CONST) > 32:
            _CONST.clear()
        h4p = torch.tensor([[0, 0, 0, H, W, H, W, 0]], dtype=torch.float32, device=dev).repeat(B, 1)
        pidx = torch.arange(H * W, dtype=torch.float32, device=dev).repeat(B, 1)
        eye = torch.eye(3, dtype=torch.float64, device=dev).repeat(B, 1, 1).contiguous()
        _CONST[key] = (h4p, pidx, eye)
    return _CONST[key]


def homo_refine(net, template: torch.Tensor, search: torch.Tensor, iterations: int = 2, cache_template: bool = True, patch_1: torch.Tensor = None,
                per_sample: bool = False):
    """template / search: [B,1,127,127] normalised gray crops on the device (get_template_info / get_search_info output).

    Returns (H_comp [B,3,3] float64 = product of the normalised inverse homographies, as the tracker composes it,
    similarity_norm, similarity_norm_simi of the LAST iteration — the values the tracker's `> 2.5` gate sees).
    ShareFeature(template) is computed once (it is constant: SURVEY §3d) — or not at all when the caller hands it in as `patch_1`
    (the tracker keeps it for the whole sequence: the template crop is cut once, in init).
    per_sample: scores of every sample ([B]) instead of sample 0's — B independent sequences (hdn_amd.batched_tracker)."""
    if iterations < 1:
        raise ValueError("iterations must be >= 1")
    B, _, H, W = template.shape
    dev = _lib.require_device(template, search)
    h4p, pidx, eye = _constants(dev, B, H, W)
    H_comp = eye.clone()
    cur = search
    if per_sample and patch_1 is None and not cache_template:
        raise ValueError("per_sample scores need the cached-template form (cache_template=True or patch_1)")
    if patch_1 is not None:
        if patch_1.shape != template.shape or patch_1.device != template.device:
            raise ValueError(f"patch_1 must be ShareFeature(template): {tuple(template.shape)} on {template.device}, got {tuple(patch_1.shape)}")
        p1 = patch_1
    else:
        p1 = net.ShareFeature(template) if cache_template else None
    score = score_simi = None
    for _ in range(iterations):
        if p1 is not None:
            H_mat, score, score_simi = track_proj_pair(net, template, cur, h4p, p1, per_sample=per_sample)
        else:
            imgs = torch.cat((template, cur), dim=1)
            data = {"org_imgs": imgs, "input_tensors": imgs, "h4p": h4p, "patch_indices": pidx}
            H_mat, score, score_simi = track_proj(net, data, None)
        cur = refine_warp(H_mat, cur, H_comp)
    return H_comp, score, score_simi
