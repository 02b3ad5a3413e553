"""CPU oracle for the HDN homography-estimation hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and there only as the checker / the timed CPU baseline.  The product
package (hdn_amd/) never imports it and has no CPU fallback.

It restates, with stock PyTorch-CPU ops (the reference *is* PyTorch: every op on
its path is an ATen call), the arithmetic of

    xcorr_depthwise            /root/reference/hdn/core/xcorr.py:37-46
    xcorr_depthwise_circular   /root/reference/hdn/core/xcorr.py:48-61
    PreShareFeature.forward    .../Oneline_DLTv1/preprocess/input_feature_extractor.py:3-29
    DLT_solve                  .../Oneline_DLTv1/utils.py:7-67
    transformer                .../Oneline_DLTv1/utils.py:70-254
    transform                  .../Oneline_DLTv1/utils.py:257-274
    track_proj (post-trunk)    /root/reference/hdn/models/model_builder_e2e_unconstrained_v2.py:161-217
    HomoModelBuilder.forward   .../Oneline_DLTv1/models/homo_model_builder.py:115-217

Parity pinning: the reference has no tests / golden vectors of its own
(SURVEY.md §4), so this oracle is pinned against outputs of the reference itself,
captured in this container by tests/golden/make_golden.py and committed as
tests/golden/*.npz (checked by tests/test_oracle_golden.py).

A float64 numpy restatement of the two correlations (`*_f64`) is included as the
rounding-free truth used to bound fp32 summation-order differences.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------- #
# correlations
# --------------------------------------------------------------------------- #
def xcorr_depthwise(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """out[b,c,i,j] = sum_{u,v} x[b,c,i+u,j+v] * kernel[b,c,u,v]  (valid, stride 1, no flip).

    Reference: hdn/core/xcorr.py:37-46 folds batch*channel into conv groups.
    """
    B, C, Hk, Wk = kernel.shape
    planes = x.reshape(1, B * C, x.shape[2], x.shape[3])
    taps = kernel.reshape(B * C, 1, Hk, Wk)
    y = F.conv2d(planes, taps, groups=B * C)
    return y.reshape(B, C, y.shape[2], y.shape[3])


def xcorr_fast(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """out[b,o,i,j] = sum_c sum_uv x[b,c,i+u,j+v] * kernel[b,o*C+c,u,v].  Reference: hdn/core/xcorr.py:26-34
    (xcorr_slow, :10-23, computes the same batch by batch)."""
    B, C = x.shape[0], x.shape[1]
    pk = kernel.reshape(-1, C, kernel.shape[2], kernel.shape[3])
    px = x.reshape(1, -1, x.shape[2], x.shape[3])
    po = F.conv2d(px, pk, groups=B)
    return po.reshape(B, -1, po.shape[2], po.shape[3])


def circular_pad_index(Hx: int, Wx: int):
    """Row / column source indices of the padded plane of xcorr.py:52-53.

    Rows (theta axis of the log-polar map) wrap around by Hx//2 on each side; then
    columns (log-rho axis) are edge-replicated by Wx//2 on each side.
    """
    ph, pw = Hx // 2, Wx // 2
    rows = torch.arange(-ph, Hx + ph) % Hx
    cols = torch.arange(-pw, Wx + pw).clamp(0, Wx - 1)
    return rows, cols


def xcorr_depthwise_circular(x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """Reference: hdn/core/xcorr.py:48-61 (circular-H pad, then replicate-W pad, then depthwise corr)."""
    rows, cols = circular_pad_index(x.shape[2], x.shape[3])
    xp = x.index_select(2, rows).index_select(3, cols)
    return xcorr_depthwise(xp, kernel)


def xcorr_depthwise_f64(x: np.ndarray, k: np.ndarray) -> np.ndarray:
    """float64 truth of xcorr_depthwise (numpy, tap-by-tap accumulation)."""
    x = np.asarray(x, np.float64)
    k = np.asarray(k, np.float64)
    B, C, Hx, Wx = x.shape
    Hk, Wk = k.shape[2:]
    Ho, Wo = Hx - Hk + 1, Wx - Wk + 1
    out = np.zeros((B, C, Ho, Wo), np.float64)
    for u in range(Hk):
        for v in range(Wk):
            out += x[:, :, u : u + Ho, v : v + Wo] * k[:, :, u : u + 1, v : v + 1]
    return out


def xcorr_depthwise_circular_f64(x: np.ndarray, k: np.ndarray) -> np.ndarray:
    rows, cols = circular_pad_index(x.shape[2], x.shape[3])
    xp = np.asarray(x, np.float64)[:, :, rows.numpy()][:, :, :, cols.numpy()]
    return xcorr_depthwise_f64(xp, k)


# --------------------------------------------------------------------------- #
# PreShareFeature
# --------------------------------------------------------------------------- #
SHARE_FEATURE_CHANNELS = (1, 4, 8, 1)
# nn.Sequential slots of the reference module: conv at 0/3/6, BatchNorm at 1/4/7
SHARE_FEATURE_SLOTS = ((0, 1), (3, 4), (6, 7))


def share_feature(x: torch.Tensor, sd: dict) -> torch.Tensor:
    """3 x (conv3x3 pad 1 no bias -> eval BatchNorm -> ReLU), channels 1->4->8->1.

    `sd` is the reference module's state_dict (keys 'ShareFeature.<slot>.<name>').
    Reference: preprocess/input_feature_extractor.py:6-18,27-29.
    """
    y = x
    for conv_slot, bn_slot in SHARE_FEATURE_SLOTS:
        p = lambda slot, n: torch.as_tensor(sd[f"ShareFeature.{slot}.{n}"])
        y = F.conv2d(y, p(conv_slot, "weight"), padding=1)
        y = F.batch_norm(
            y, p(bn_slot, "running_mean"), p(bn_slot, "running_var"), p(bn_slot, "weight"), p(bn_slot, "bias"),
            training=False, eps=BN_EPS,
        )
        y = F.relu(y)
    return y


# --------------------------------------------------------------------------- #
# DLT
# --------------------------------------------------------------------------- #
# utils.py:18-26 with divide == 1 gathers columns [0,1,2,3,6,7,4,5]: the caller's corner
# order TL,BL,BR,TR (get_img_info.py:93-98) becomes TL,BL,TR,BR inside the solve.
DLT_POINT_ORDER = (0, 1, 3, 2)


def dlt_system(src_p: torch.Tensor, off_set: torch.Tensor):
    """The 8x8 system A h = b of utils.py:38-60 for every sample.  src_p, off_set: [B, 8]."""
    B = src_p.shape[0]
    src = src_p.reshape(B, 4, 2)[:, DLT_POINT_ORDER, :]
    dst = src + off_set.reshape(B, 4, 2)[:, DLT_POINT_ORDER, :]
    x, y = src[..., 0], src[..., 1]
    u, v = dst[..., 0], dst[..., 1]
    one, zero = torch.ones_like(x), torch.zeros_like(x)
    row_u = torch.stack([x, y, one, zero, zero, zero, -(u * x), -(u * y)], dim=-1)
    row_v = torch.stack([zero, zero, zero, x, y, one, -(v * x), -(v * y)], dim=-1)
    A = torch.stack([row_u, row_v], dim=2).reshape(B, 8, 8)
    b = torch.stack([u, v], dim=2).reshape(B, 8, 1)
    return A, b


def dlt_solve(src_p: torch.Tensor, off_set: torch.Tensor) -> torch.Tensor:
    """4-point DLT: H maps src -> src+off.  Returns [B, 1, 3, 3] with H[2,2] == 1.

    Reference: utils.py:7-67 (h8 = inverse(A) @ b, then append 1).
    """
    A, b = dlt_system(src_p, off_set)
    h8 = torch.matmul(torch.inverse(A), b).reshape(-1, 8)
    H = torch.cat([h8, torch.ones_like(h8[:, :1])], dim=1)
    return H.reshape(-1, 1, 3, 3)


def dlt_solve_f64(src_p: np.ndarray, off_set: np.ndarray) -> np.ndarray:
    A, b = dlt_system(torch.as_tensor(src_p, dtype=torch.float64), torch.as_tensor(off_set, dtype=torch.float64))
    h8 = torch.linalg.solve(A, b).reshape(-1, 8)
    return torch.cat([h8, torch.ones_like(h8[:, :1])], dim=1).reshape(-1, 3, 3).numpy()


# --------------------------------------------------------------------------- #
# projective spatial transformer
# --------------------------------------------------------------------------- #
def sampling_grid(height: int, width: int) -> torch.Tensor:
    """[3, H*W] homogeneous grid of utils.py:192-213: x,y in linspace(-1,1,.) , row-major (y outer)."""
    xs = torch.linspace(-1.0, 1.0, width)
    ys = torch.linspace(-1.0, 1.0, height)
    gx = xs.unsqueeze(0).expand(height, width).reshape(-1)
    gy = ys.unsqueeze(1).expand(height, width).reshape(-1)
    return torch.stack([gx, gy, torch.ones_like(gx)], dim=0)


def transformer(U: torch.Tensor, theta: torch.Tensor, out_size):
    """Projective bilinear sampler.  U: [B,C,H,W] (the reference passes NCHW despite its docstring),
    theta: [B,3,3] acting on the [-1,1]^2 grid.  Returns ([B,H,W,C], condition).

    Reference: utils.py:70-254.  Semantics reproduced, not "fixed":
      * x = (x_s + 1) * W / 2  (so the grid spans [0, W], not [0, W-1])   utils.py:128-129
      * x0 = floor(x), x1 = x0 + 1, each clamped to [0, W-1]               utils.py:132-140
      * weights use the CLAMPED integer taps against the UNCLAMPED x       utils.py:181-184
        => if both taps clamp to the same index the pixel is exactly 0
      * t += 1e-6 where |t| < 1e-7                                          utils.py:236-240
    """
    B, C, H, W = U.shape
    oh, ow = out_size
    assert (oh, ow) == (H, W), "the reference's index expand (utils.py:164) requires out_size == input size"
    grid = sampling_grid(oh, ow)
    T = torch.matmul(theta.reshape(-1, 3, 3).float(), grid.unsqueeze(0).expand(B, 3, oh * ow))
    t = T[:, 2, :].reshape(-1)
    t = t + 1e-6 * (1.0 - (t.abs() >= 1e-7).float())
    condition = (t.abs() > 1e-7).float().sum()
    xs = T[:, 0, :].reshape(-1) / t
    ys = T[:, 1, :].reshape(-1) / t

    x = (xs + 1.0) * W / 2.0
    y = (ys + 1.0) * H / 2.0
    x0 = torch.floor(x).int()
    y0 = torch.floor(y).int()
    x1 = x0 + 1
    y1 = y0 + 1
    x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
    y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)

    flat = U.permute(0, 2, 3, 1).reshape(-1, C).float()
    base = (torch.arange(B) * (H * W)).repeat_interleave(oh * ow)
    take = lambda yy, xx: flat[(base + yy.long() * W + xx.long())]
    Ia, Ib, Ic, Id = take(y0, x0), take(y1, x0), take(y0, x1), take(y1, x1)
    x0f, x1f, y0f, y1f = x0.float(), x1.float(), y0.float(), y1.float()
    wa = ((x1f - x) * (y1f - y)).unsqueeze(1)
    wb = ((x1f - x) * (y - y0f)).unsqueeze(1)
    wc = ((x - x0f) * (y1f - y)).unsqueeze(1)
    wd = ((x - x0f) * (y - y0f)).unsqueeze(1)
    out = wa * Ia + wb * Ib + wc * Ic + wd * Id
    return out.reshape(B, oh, ow, C), condition


def norm_matrices(batch: int, half: float = 63.5):
    """M and inverse(M) of model_builder_e2e_unconstrained_v2.py:196-209 (w_h_scala = 63.5)."""
    M = torch.tensor([[half, 0.0, half], [0.0, half, half], [0.0, 0.0, 1.0]])
    Minv = torch.inverse(M)
    return M.unsqueeze(0).expand(batch, 3, 3), Minv.unsqueeze(0).expand(batch, 3, 3)


def transform(patch_h, patch_w, M_inv, H_mat, M, I1, patch_indices, batch_base) -> torch.Tensor:
    """Warp I1 by H' = M_inv @ H @ M, then gather the patch pixels.  Reference: utils.py:257-274."""
    B, C, H, W = I1.shape
    Hn = torch.matmul(torch.matmul(M_inv, H_mat), M)
    warped, _ = transformer(I1, Hn, (H, W))
    flat = warped.reshape(-1, C)
    pix = patch_indices.reshape(-1).long() + batch_base
    return flat[pix].reshape(B, patch_h, patch_w, C).permute(0, 3, 1, 2)


def full_patch_indices(B: int, H: int, W: int):
    """patch_indices / batch base as built by get_img_info.py:88-92 and model_builder…:169-172."""
    pidx = torch.arange(H * W, dtype=torch.float32).unsqueeze(0).expand(B, -1)
    base = (torch.arange(B) * (H * W)).unsqueeze(1).expand(B, H * W).reshape(-1)
    return pidx, base


def dlt_warp(h4p: torch.Tensor, off: torch.Tensor, img: torch.Tensor):
    """DLT_solve + transform over the full HxW patch: the fused stage the HIP kernel implements.

    Returns (H_mat [B,3,3], warped [B,1,H,W]).
    """
    B, _, H, W = img.shape
    Hm = dlt_solve(h4p, off).squeeze(1)
    M, Minv = norm_matrices(B, W / 2.0)
    pidx, base = full_patch_indices(B, H, W)
    return Hm, transform(H, W, Minv, Hm, M, img, pidx, base)


# --------------------------------------------------------------------------- #
# host-side input prep (get_img_info.py)
# --------------------------------------------------------------------------- #
MEAN_BGR = np.array([118.93, 113.97, 102.60])
STD_BGR = np.array([69.85, 68.81, 72.45])


def gray_normalise(crop_hwc: np.ndarray) -> np.ndarray:
    """HxWx3 (uint8-valued) -> 1xHxW float64: mean over channels of (x-mean)/std.  get_img_info.py:15-36."""
    z = (np.asarray(crop_hwc, np.float64) - MEAN_BGR.reshape(1, 1, 3)) / STD_BGR.reshape(1, 1, 3)
    return np.transpose(z.mean(axis=2, keepdims=True), (2, 0, 1))


def merge_pair(tmp: np.ndarray, search: np.ndarray) -> dict:
    """get_img_info.py:72-103 for the 127x127 full-patch case."""
    org = np.concatenate([tmp, search], axis=0)
    Hh, Ww = org.shape[1:]
    yy, xx = np.meshgrid(np.arange(Hh), np.arange(Ww), indexing="ij")
    return {
        "org_imgs": org,
        "input_tensors": org.copy(),
        "patch_indices": (yy * Ww + xx).reshape(-1).astype(np.float64),
        "four_points": np.array([0, 0, 0, Hh, Ww, Hh, Ww, 0], np.float64),
    }


# --------------------------------------------------------------------------- #
# track_proj / HomoModelBuilder.forward around a caller-supplied trunk
# --------------------------------------------------------------------------- #
def track_proj(data: dict, sf_sd: dict, regress):
    """model_builder_e2e_unconstrained_v2.py:161-217.  `regress(x[B,2,H,W]) -> [B,8]` is the trunk+avgpool+fc."""
    org, inp, h4p = data["org_imgs"], data["input_tensors"], data["h4p"]
    p1 = share_feature(inp[:, :1], sf_sd)
    p2 = share_feature(inp[:, 1:], sf_sd)
    x = regress(torch.cat([p1, p2], dim=1))
    Hm, pred = dlt_warp(h4p, x, org[:, :1])
    pf = share_feature(pred, sf_sd)
    n = float(inp.shape[2] * inp.shape[3])
    score = (p2 - pf).abs()[0][0].sum() / (127 * 127)
    score_simi = (p2 - p1).abs()[0][0].sum() / (127 * 127)
    return Hm, score, score_simi, {"x": x, "pred_I2": pred, "patch_1": p1, "patch_2": p2, "pred_feat": pf, "n": n}


def homo_forward(data: dict, sf_sd: dict, regress) -> dict:
    """HomoModelBuilder.forward (homo_model_builder.py:115-217) with default if_pos / if_unsup (all ones)."""
    Hm, _, _, aux = track_proj(data, sf_sd, regress)
    p1, p2, pf, x, pred = aux["patch_1"], aux["patch_2"], aux["pred_feat"], aux["x"], aux["pred_I2"]
    B = x.shape[0]
    # TripletMarginLoss(margin=1, p=1, reduce=False)(anchor=search, positive=pred, negative=template)
    # on [B,1,127,127] tensors: pairwise_distance is over the LAST dim, eps=1e-6 added to the difference
    d_ap = ((p2 - pf) + 1e-6).abs().sum(-1)
    d_an = ((p2 - p1) + 1e-6).abs().sum(-1)
    loss_mat = (d_ap - d_an + 1.0).clamp_min(0.0)
    n_pos = B * 1 * 127 * 127  # pos_ids = nonzero rows of an all-ones [B,1,127,127] tensor
    feature_loss = (loss_mat.sum() / n_pos / (127 * 127)).reshape(1)
    return {
        "feature_loss": feature_loss,
        "pred_I2_d": pred[:1],
        "x": x,
        "H_mat": Hm,
        "patch_2_res_d": p2[:1],
        "pred_I2_CnnFeature_d": pf[:1],
        "homo_neg_loss": torch.tensor(0.0),
    }


# --------------------------------------------------------------------------- #
# the tracker's refinement loop (hdn/tracker/hdn_tracker_proj_e2e.py:242-250) with OpenCV's warp restated
# --------------------------------------------------------------------------- #
def warp_perspective_replicate(img: np.ndarray, M: np.ndarray) -> np.ndarray:
    """cv2.warpPerspective(img, M, (W, H), flags=INTER_LINEAR, borderMode=BORDER_REPLICATE) for one 2-D image.

    PARITY UNPINNED.  cv2 (opencv-python, unpinned in the reference's INSTALL.md) is absent from the reference tree and from
    this image, so this is a restatement of OpenCV 4.x's published algorithm (modules/imgproc/src/imgwarp.cpp:
    WarpPerspectiveInvoker and remapBilinear), not something checked against cv2 output:
      M is inverted in float64 (dst(x, y) = src(M^-1 (x, y, 1))); destination pixels are walked in blocks of
      bw x bh = 64 x 16 (for 127 x 127); per pixel  W = W0 + M6*x1, W = 32/W (0 if W == 0),
      X = cvRound((X0 + M0*x1) * W), Y likewise, with X0 = M0*bx + M1*y + M2 etc. evaluated in that order;
      sx = X >> 5, fx = (X & 31) / 32; weights (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx as float32 products; taps clamped
      to the image; sum left to right in the image's type."""
    img = np.asarray(img)
    Hh, Ww = img.shape
    m = _inv3(np.asarray(M, np.float64))
    bh = min(16, Hh)
    bw = min(1024 // bh, Ww)
    bh = min(1024 // bw, Hh)
    y = np.arange(Hh, dtype=np.float64)[:, None]
    x = np.arange(Ww)[None, :]
    bx = ((x // bw) * bw).astype(np.float64)
    x1 = (x - (x // bw) * bw).astype(np.float64)
    X0 = m[0, 0] * bx + m[0, 1] * y + m[0, 2]
    Y0 = m[1, 0] * bx + m[1, 1] * y + m[1, 2]
    W0 = m[2, 0] * bx + m[2, 1] * y + m[2, 2]
    Wd = W0 + m[2, 0] * x1
    with np.errstate(divide="ignore"):
        Wd = np.where(Wd != 0.0, 32.0 / np.where(Wd != 0.0, Wd, 1.0), 0.0)
    fX = np.clip((X0 + m[0, 0] * x1) * Wd, -2147483648.0, 2147483647.0)
    fY = np.clip((Y0 + m[1, 0] * x1) * Wd, -2147483648.0, 2147483647.0)
    X = np.rint(fX).astype(np.int64)   # cvRound: half to even
    Y = np.rint(fY).astype(np.int64)
    sx, sy = X >> 5, Y >> 5
    fx = ((X & 31).astype(np.float32)) * np.float32(1.0 / 32.0)
    fy = ((Y & 31).astype(np.float32)) * np.float32(1.0 / 32.0)
    one = np.float32(1.0)
    w00, w01, w10, w11 = (one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx
    x0, x1c = np.clip(sx, 0, Ww - 1), np.clip(sx + 1, 0, Ww - 1)
    y0, y1c = np.clip(sy, 0, Hh - 1), np.clip(sy + 1, 0, Hh - 1)
    wt = img.dtype.type
    return ((img[y0, x0] * w00.astype(wt) + img[y0, x1c] * w01.astype(wt)) + img[y1c, x0] * w10.astype(wt)) + img[y1c, x1c] * w11.astype(wt)


def _inv3(m: np.ndarray) -> np.ndarray:
    """3x3 inverse by adjugate / determinant in float64 (cv::invert's 3x3 case)."""
    a, b, c, d, e, f, g, h, i = m.reshape(-1)
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    idet = 1.0 / det if det != 0.0 else 0.0
    return np.array([[(e * i - f * h), (c * h - b * i), (b * f - c * e)],
                     [(f * g - d * i), (a * i - c * g), (c * d - a * f)],
                     [(d * h - e * g), (b * g - a * h), (a * e - b * d)]], np.float64) * idet


def refine_step(H: np.ndarray, search: np.ndarray, H_comp: np.ndarray):
    """hdn_tracker_proj_e2e.py:246-250 for one sample: H [3,3] float32 (track_proj's H_mat), search [127,127] float32,
    H_comp [3,3] float64 -> (warped search, H_comp @ H_hm).  The float32 inverses follow numpy's dtype rules there
    (np.linalg.inv of a float32 array is float32); their last-bit rounding depends on LAPACK's sgesv and is not reproduced
    beyond float32 precision."""
    t = _inv3(np.asarray(H, np.float32).astype(np.float64)).astype(np.float32)
    Hhm = ((1.0 / np.float64(t[2, 2])) * t.astype(np.float64)).astype(np.float32)
    M = _inv3(Hhm.astype(np.float64)).astype(np.float32)
    warped = warp_perspective_replicate(np.asarray(search, np.float32), M.astype(np.float64))
    return warped.astype(np.float32), np.asarray(H_comp, np.float64) @ Hhm.astype(np.float64)


def homo_refine(template: torch.Tensor, search: torch.Tensor, sf_sd: dict, regress, iterations: int = 2):
    """The refinement loop of hdnTrackerHomo.track_new (:242-250) with trip count `iterations` for a batch of independent
    pairs: template / search [B,1,127,127] normalised gray crops.  Returns (H_comp [B,3,3] float64, scores of the last
    iteration, list of per-iteration H_mat)."""
    B = template.shape[0]
    H_comp = np.tile(np.eye(3), (B, 1, 1))
    cur = search.clone()
    h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(B, 1)
    Hs, score, score_simi = [], None, None
    for _ in range(iterations):
        imgs = torch.cat([template, cur], dim=1)
        Hm, score, score_simi, _ = track_proj({"org_imgs": imgs, "input_tensors": imgs, "h4p": h4p}, sf_sd, regress)
        Hs.append(Hm)
        nxt = []
        for b in range(B):
            w, H_comp[b] = refine_step(Hm[b].numpy(), cur[b, 0].numpy(), H_comp[b])
            nxt.append(torch.from_numpy(w))
        cur = torch.stack(nxt).unsqueeze(1)
    return H_comp, score, score_simi, Hs


def corner_error(pred_off: np.ndarray, ref_off: np.ndarray) -> np.ndarray:
    """sqrt(sum ||delta||^2 / 4) per sample — toolkit/utils/statistics.py:206-218 (success_4pts_error)."""
    d = (np.asarray(pred_off, np.float64) - np.asarray(ref_off, np.float64)).reshape(-1, 4, 2)
    return np.sqrt((d ** 2).sum(axis=(1, 2)) / 4.0)


# --------------------------------------------------------------------------- #
# log-polar resample
# --------------------------------------------------------------------------- #
def logpolar_tables(size: int, rot: float = 0.0):
    """The 1-D factors of STN_Polar._prepare_grid (hdn/models/logpolar.py:58-74) for an output of size x size:
    rho[b] = exp(b * log(size/2)/size) - 1,  cos/sin(theta[a]) with theta[a] = a * 2*pi/size + rot."""
    ls = torch.linspace(0, size - 1, size)
    mag = math.log(size / 2) / size
    rho = torch.exp(mag * ls) - 1.0
    theta = ls * 2.0 * math.pi / size + rot
    return rho, torch.cos(theta), torch.sin(theta)


def logpolar_sample(x: torch.Tensor, polar: torch.Tensor, delta=(0, 0), image_sz: int = None):
    """STN_Polar(image_sz).forward(x, polar, delta) -> (x_lp [B,C,S,S], grid [B,S,S,2]), S = image_sz//2; image_sz
    defaults to the crop's width (the reference also applies STN_Polar(255) to 127-px crops: update_template).

    Reference: hdn/models/logpolar.py:100-124: grid = (rho*cos(theta) + polar_x, rho*sin(theta) + polar_y) / (size//2),
    then F.grid_sample(bilinear, padding_mode='border', align_corners=False)."""
    B, C, H, W = x.shape
    S = (W if image_sz is None else image_sz) // 2
    rho, c, s = logpolar_tables(S, float(delta[1]))
    ix = rho.unsqueeze(0) * c.unsqueeze(1)   # [a][b] = rho[b] * cos(theta[a])
    iy = rho.unsqueeze(0) * s.unsqueeze(1)
    gx = (ix.unsqueeze(0) + polar[:, 0].reshape(B, 1, 1)) / (H // 2)
    gy = (iy.unsqueeze(0) + polar[:, 1].reshape(B, 1, 1)) / (W // 2)
    grid = torch.stack([gx, gy], dim=3)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="border", align_corners=False), grid


# --------------------------------------------------------------------------- #
# correlation heads: MultiBAN / MultiCircBAN with a reference state_dict
# --------------------------------------------------------------------------- #
def _seq_conv_bn_relu(x, sd, prefix):
    y = F.conv2d(x, sd[prefix + ".0.weight"])
    y = F.batch_norm(y, sd[prefix + ".1.running_mean"], sd[prefix + ".1.running_var"], sd[prefix + ".1.weight"],
                     sd[prefix + ".1.bias"], training=False, eps=BN_EPS)
    return F.relu(y)


def multi_ban(z_fs, x_fs, sd: dict, circular: bool):
    """MultiBAN.forward (hdn/models/head/ban.py:102-127) / MultiCircBAN.forward (ban_lp.py:66-92), weighted=True.

    Per level: cls/loc = head(xcorr(conv_search(x), conv_kernel(z))); loc *= loc_scale[i]; softmax-weighted sums."""
    corr = xcorr_depthwise_circular if circular else xcorr_depthwise
    cls, loc = [], []
    for i, (z, x) in enumerate(zip(z_fs, x_fs)):
        outs = []
        for br in ("cls", "loc"):
            p = f"box{i + 2}.{br}"
            f = corr(_seq_conv_bn_relu(x, sd, p + ".conv_search"), _seq_conv_bn_relu(z, sd, p + ".conv_kernel"))
            h = _seq_conv_bn_relu(f, sd, p + ".head")  # head.0 conv1x1, head.1 BN, head.2 ReLU
            outs.append(F.conv2d(h, sd[p + ".head.3.weight"], sd[p + ".head.3.bias"]))
        cls.append(outs[0])
        loc.append(outs[1] * sd["loc_scale"][i])
    cw, lw = F.softmax(sd["cls_weight"], 0), F.softmax(sd["loc_weight"], 0)
    c = sum(cls[i] * cw[i] for i in range(len(cls)))
    l = sum(loc[i] * lw[i] for i in range(len(loc)))
    return c, l
