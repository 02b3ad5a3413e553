"""Drop-ins for the Oneline_DLTv1 geometry helpers, on HIP kernels.

    DLT_solve(src_p, off_set)                  <- homo_estimator/Deep_homography/Oneline_DLTv1/utils.py:7-67
    transformer(U, theta, out_size)            <- utils.py:70-254
    transform(ph, pw, M_inv, H, M, I1, pidx, base)   <- utils.py:257-274   (alias Homo_STN)
    dlt_warp(h4p, off, img)                    fused DLT_solve + transform for the full-patch case

Signatures, argument meaning and output shapes follow the reference.
"""
from __future__ import annotations

import torch

from . import _lib


def DLT_solve(src_p: torch.Tensor, off_set: torch.Tensor) -> torch.Tensor:
    """4-point DLT.  src_p, off_set: [B, 8] -> H [B, 1, 3, 3] mapping src -> src + off (H[2,2] = 1).

    Like the reference (utils.py:12: divide = int(sqrt(len/2) - 1)) only the single-quad case (8 values
    per sample) is meaningful; other lengths are rejected.
    """
    if src_p.dim() != 2 or src_p.shape[1] != 8 or off_set.shape != src_p.shape:
        raise ValueError(f"DLT_solve expects [B,8] corner / offset tensors, got {tuple(src_p.shape)} and {tuple(off_set.shape)}")
    dev = _lib.require_device(src_p, off_set)
    s, o = src_p.detach().contiguous(), off_set.detach().contiguous()
    B = s.shape[0]
    if B == 0:
        raise ValueError("empty batch")
    H = torch.empty((B, 9), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_dlt_solve_f32(_lib.ptr(s), _lib.ptr(o), _lib.ptr(H), B, _lib.stream_ptr(dev))
    _lib.check(rc, "DLT_solve")
    return H.view(B, 1, 3, 3)


def transformer(U: torch.Tensor, theta: torch.Tensor, out_size, want_condition: bool = True, **kwargs):
    """Projective spatial transformer: U [B,C,H,W], theta [B,3,3] (or [B,9]) -> ([B,H,W,C], condition).

    `condition` (the count of |t| > 1e-7 after the nudge, utils.py:241; unused by every caller in the reference)
    is returned as a 0-dim fp32 device tensor like the reference's; it costs one atomic per wave.
    """
    if U.dim() != 4:
        raise ValueError(f"U must be [B,C,H,W], got {tuple(U.shape)}")
    B, C, H, W = U.shape
    if tuple(out_size) != (H, W):
        # the reference's idx.expand(height*width*num_batch, C) (utils.py:164) only works in this case
        raise ValueError(f"out_size {tuple(out_size)} must equal the input size {(H, W)} (as in the reference)")
    dev = _lib.require_device(U, theta)
    th = theta.detach().reshape(-1, 9).contiguous()
    if th.shape[0] != B:
        raise ValueError(f"theta batch {th.shape[0]} != image batch {B}")
    img = U.detach().contiguous()
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev) if want_condition else None  # zeroed by the call
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_warp_count_f32(_lib.ptr(img), _lib.ptr(th), _lib.ptr(out),
                                            _lib.ptr(count) if want_condition else None, B, C, H, W, _lib.stream_ptr(dev))
    _lib.check(rc, "transformer")
    return out, (count[0].to(torch.float32) if want_condition else None)


def _is_full_patch(patch_indices: torch.Tensor, B: int, ph: int, pw: int, H: int, W: int) -> bool:
    return (ph, pw) == (H, W) and tuple(patch_indices.shape) == (B, H * W)


def transform(patch_size_h, patch_size_w, M_tile_inv, H_mat, M_tile, I1, patch_indices, batch_indices_tensor,
              assume_identity_patch: bool = False):
    """Warp I1 by M_inv @ H @ M and gather the patch pixels -> [B, C, ph, pw].

    With `assume_identity_patch=True` the final gather (utils.py:268-272) is skipped; that is exact when
    patch_indices is the row-major index of the full image (get_img_info.py:88-92), which is how the
    tracker always calls it.
    """
    B, C, H, W = I1.shape
    dev = _lib.require_device(I1, H_mat)
    Hn = torch.matmul(torch.matmul(M_tile_inv.to(dev), H_mat), M_tile.to(dev))
    warped, _ = transformer(I1, Hn, (H, W), want_condition=False)
    if assume_identity_patch and _is_full_patch(patch_indices, B, patch_size_h, patch_size_w, H, W):
        return warped.permute(0, 3, 1, 2)
    flat = warped.reshape(-1, C)
    pix = patch_indices.reshape(-1).long().to(dev) + batch_indices_tensor.to(dev)
    return flat[pix].reshape(B, patch_size_h, patch_size_w, C).permute(0, 3, 1, 2)


Homo_STN = transform  # the name model_builder_e2e_unconstrained_v2.py:30 imports it under


def dlt_warp(h4p: torch.Tensor, off_set: torch.Tensor, img: torch.Tensor):
    """Fused DLT_solve + transform (full patch, M = [[W/2,0,W/2],[0,H/2,H/2],[0,0,1]]).

    h4p, off_set: [B,8]; img: [B,1,H,W] -> (H_mat [B,3,3], warped [B,1,H,W]).
    Replaces model_builder_e2e_unconstrained_v2.py:195-210 / homo_model_builder.py:166-170.
    """
    if img.dim() != 4 or img.shape[1] != 1:
        raise ValueError(f"img must be [B,1,H,W], got {tuple(img.shape)}")
    B, _, H, W = img.shape
    if tuple(h4p.shape) != (B, 8) or tuple(off_set.shape) != (B, 8):
        raise ValueError(f"h4p / off_set must be [{B},8], got {tuple(h4p.shape)} and {tuple(off_set.shape)}")
    dev = _lib.require_device(h4p, off_set, img)
    p, o, im = h4p.detach().contiguous(), off_set.detach().contiguous(), img.detach()
    # one channel of a [B,C,H,W] tensor (HomoModelBuilder.forward: org_imgs[:, :1]) is read in place: rows and pixels contiguous, images C * H * W apart
    if not (im.stride(3) == 1 and im.stride(2) == W) or (B > 1 and im.stride(0) < H * W):
        im = im.contiguous()
    bstride = im.stride(0) if B > 1 else H * W
    Hm = torch.empty((B, 9), dtype=torch.float32, device=dev)
    warped = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_dlt_warp_strided_f32(_lib.ptr(p), _lib.ptr(o), _lib.ptr(im), bstride, _lib.ptr(Hm), _lib.ptr(warped), B, H, W,
                                                  _lib.stream_ptr(dev))
    _lib.check(rc, "dlt_warp")
    return Hm.view(B, 3, 3), warped


def l1_score(a: torch.Tensor, b: torch.Tensor, scale: float) -> torch.Tensor:
    """sum |a - b| * scale as a 0-dim device tensor (track_proj's similarity scores)."""
    dev = _lib.require_device(a, b)
    if a.numel() != b.numel() or a.numel() == 0:
        raise ValueError("l1_score needs two non-empty tensors of equal size")
    ac, bc = a.detach().contiguous(), b.detach().contiguous()
    out = torch.empty((1,), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_l1_score_f32(_lib.ptr(ac), _lib.ptr(bc), _lib.ptr(out), ac.numel(), float(scale), _lib.stream_ptr(dev))
    _lib.check(rc, "l1_score")
    return out[0]


def l1_score2(a: torch.Tensor, b0: torch.Tensor, b1: torch.Tensor, scale: float):
    """(sum |a - b0|, sum |a - b1|) * scale as two 0-dim device tensors from ONE launch: track_proj's two scores share
    their first operand (model_builder_e2e_unconstrained_v2.py:213-216).  Bit-identical to two l1_score calls."""
    dev = _lib.require_device(a, b0, b1)
    if a.numel() != b0.numel() or a.numel() != b1.numel() or a.numel() == 0:
        raise ValueError("l1_score2 needs three non-empty tensors of equal size")
    ac, b0c, b1c = a.detach().contiguous(), b0.detach().contiguous(), b1.detach().contiguous()
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_l1_score2_f32(_lib.ptr(ac), _lib.ptr(b0c), _lib.ptr(b1c), _lib.ptr(out), ac.numel(), float(scale),
                                           _lib.stream_ptr(dev))
    _lib.check(rc, "l1_score2")
    return out[0], out[1]


def l1_score2_batch(a: torch.Tensor, b0: torch.Tensor, b1: torch.Tensor, scale: float):
    """Per-sample scores of a batch: a, b0, b1 [B,1,H,W] (channel 0 of every sample, as track_proj scores it) -> two [B] device tensors
    from ONE launch (hdn_l1_score2_batch_f32); entry s is bit-identical to l1_score2 on sample s alone.  The reference scores sample 0
    only (`[0][0]`, model_builder_e2e_unconstrained_v2.py:213-216): its tracker holds one sequence.  The lock-step multi-sequence
    tracker needs every sequence's own gate value."""
    dev = _lib.require_device(a, b0, b1)
    if a.dim() != 4 or a.shape[1] != 1 or a.shape != b0.shape or a.shape != b1.shape or a.numel() == 0:
        raise ValueError("l1_score2_batch needs three [B,1,H,W] tensors of equal shape")
    B, n = a.shape[0], a.shape[2] * a.shape[3]
    ac, b0c, b1c = a.detach().contiguous(), b0.detach().contiguous(), b1.detach().contiguous()
    out = torch.empty((B, 2), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_l1_score2_batch_f32(_lib.ptr(ac), _lib.ptr(b0c), _lib.ptr(b1c), _lib.ptr(out), n, n, B, float(scale), _lib.stream_ptr(dev))
    _lib.check(rc, "l1_score2_batch")
    return out[:, 0], out[:, 1]

