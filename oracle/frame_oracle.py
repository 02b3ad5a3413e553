"""TEST INFRASTRUCTURE ONLY (see oracle/hdn_oracle.py): CPU restatement of the tracker's per-frame image handling, the
"device-resident frame pipeline" row of SURVEY.md §8f-3.

Pinned (held to fixtures produced by the reference's own functions, tests/golden/frame.npz):
    get_subwindow / get_subwindow_for_homo WITHOUT the resize  <- hdn/tracker/base_tracker.py:61-213 (crop, uint8 mean padding)
    get_search_info / get_template_info on 127-px crops        <- .../Oneline_DLTv1/tools/get_img_info.py:8-70
PARITY UNPINNED (OpenCV is a third-party dependency that neither the reference tree nor this image contains; these restate
OpenCV 4.x's published 8-bit algorithms from modules/imgproc/src/{resize,imgwarp}.cpp and are checked through properties
only — identity, integer shifts, borders, monotonicity):
    resize_linear_u8          cv2.resize(INTER_LINEAR), called by get_subwindow when original_sz != model_sz
    warp_perspective_u8       cv2.warpPerspective(INTER_LINEAR, BORDER_REPLICATE) of the full frame, hdn_tracker_proj_e2e.py:154
    warp_affine_cubic_u8      cv2.warpAffine(flags=2 = INTER_CUBIC, BORDER_REPLICATE), hdn/utils/transform.py:98-99
    warp_affine_linear_f32    cv2.warpAffine(float32 image, default flags = INTER_LINEAR, BORDER_CONSTANT 0), hdn/utils/transform.py:237
                              (get_mask_window; ModelBuilder.track_proj ignores the mask it is handed, so nothing downstream reads it)
    log_polar_maps / remap_linear_u8 / get_polar_img   cv2.logPolar, hdn/models/logpolar.py:11-29
"""
from __future__ import annotations

import math

import numpy as np

MEAN_I = np.array([118.93, 113.97, 102.60])
STD_I = np.array([69.85, 68.81, 72.45])


# --------------------------------------------------------------------------- pinned part
def crop_bounds(pos, original_sz, im_h, im_w):
    """base_tracker.py:75-92.  Returns (xmin, ymin, P, pads) of the UNPADDED frame: the patch is P x P with its top-left
    pixel at frame coordinate (xmin, ymin) (possibly negative), P = number of rows the reference's slice yields."""
    sz = float(original_sz)
    c = (sz - 1) / 2
    xmin = np.floor(pos[0] - c + 0.5)
    ymin = np.floor(pos[1] - c + 0.5)
    xmax, ymax = xmin + sz - 1, ymin + sz - 1
    left, top = int(max(0., -xmin)), int(max(0., -ymin))
    right, bottom = int(max(0., xmax - im_w + 1)), int(max(0., ymax - im_h + 1))
    P = int(xmax + left + 1) - int(xmin + left)
    return int(xmin), int(ymin), P, (left, top, right, bottom), (xmin + left, ymin + top, xmax + left + 1, ymax + top + 1)


def subwindow_patch(im: np.ndarray, pos, original_sz, avg_chans) -> np.ndarray:
    """The uint8 patch of get_subwindow before any resize: frame pixels inside, uint8(avg_chans) outside (numpy's
    float -> uint8 assignment truncates)."""
    H, W, C = im.shape
    xmin, ymin, P, _, _ = crop_bounds(pos, original_sz, H, W)
    ys, xs = ymin + np.arange(P), xmin + np.arange(P)
    inside = ((ys >= 0) & (ys < H))[:, None] & ((xs >= 0) & (xs < W))[None, :]
    patch = im[np.clip(ys, 0, H - 1)[:, None], np.clip(xs, 0, W - 1)[None, :], :]
    fill = np.asarray(avg_chans, np.float64).astype(np.uint8)
    return np.where(inside[:, :, None], patch, fill[None, None, :])


def get_subwindow(im, pos, model_sz, original_sz, avg_chans):
    """-> float32 [1, C, model_sz, model_sz] (base_tracker.py:61-136, islog=False)."""
    patch = subwindow_patch(im, pos, original_sz, avg_chans)
    if not np.array_equal(model_sz, original_sz):
        patch = resize_linear_u8(patch, model_sz, model_sz)
    return patch.transpose(2, 0, 1)[None].astype(np.float32)


def get_subwindow_for_homo(im, pos, model_sz, original_sz, avg_chans):
    """base_tracker.py:138-213: the same crop plus (context_xmin, context_ymin, context_xmax + 1, context_ymax + 1) in the
    padded frame's coordinates."""
    H, W, _ = im.shape
    pts = crop_bounds(pos, original_sz, H, W)[4]
    return get_subwindow(im, pos, model_sz, original_sz, avg_chans), pts


def search_info(crop_chw: np.ndarray) -> np.ndarray:
    """get_search_info / get_template_info on a [3,127,127] float32 crop -> [1,127,127] float64 normalised gray."""
    x = crop_chw.transpose(1, 2, 0).astype(np.float32)
    x = (x - MEAN_I.reshape(1, 1, 3)) / STD_I.reshape(1, 1, 3)
    return np.transpose(np.mean(x, axis=2, keepdims=True), [2, 0, 1])


# --------------------------------------------------------------------------- OpenCV restatements (unpinned)
def _cv_round(v):
    return np.rint(v).astype(np.int64)


def resize_linear_u8(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(img, (dw, dh)) for uint8, INTER_LINEAR: 11-bit fixed-point weights, HResizeLinear then
    VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>:  dst = (((b0*(S0 >> 4)) >> 16) + ((b1*(S1 >> 4)) >> 16) + 2) >> 2."""
    sh, sw, C = img.shape

    def axis(dn, sn):
        scale = 1.0 / (dn / sn)
        f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= sn - 1
        f[hi], s[hi] = 0, sn - 1
        w0 = np.clip(_cv_round((np.float32(1.0) - f) * np.float32(2048)), -32768, 32767)
        w1 = np.clip(_cv_round(f * np.float32(2048)), -32768, 32767)
        return s, w0, w1

    sx, a0, a1 = axis(dw, sw)
    sy, b0, b1 = axis(dh, sh)
    src = img.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    rows = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]     # [sh, dw, C], scaled by 2048
    S0, S1 = rows[sy], rows[np.minimum(sy + 1, sh - 1)]
    out = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def log_polar_maps(w: int, h: int, center, M: float):
    """The float32 maps cv2.logPolar(src, center, M, flags) hands to remap: cv::logPolar -> cv::warpPolar(src, size(src), center,
    maxRadius = exp(w / M), flags | WARP_POLAR_LOG) (OpenCV >= 3.4.2 / 4.x, imgproc/src/imgwarp.cpp): rho along x with
    rhos[rho] = float(exp(rho * log(maxRadius) / w) - 1), angle along y with 2 pi / h per row, x = rhos * cos + cx in double."""
    max_radius = math.exp(w / M) if M > 0 else 1.0
    kangle, kmag = 2.0 * math.pi / h, math.log(max_radius) / w
    mx = np.empty((h, w), np.float32)
    my = np.empty((h, w), np.float32)
    rhos = [float(np.float32(math.exp(rho * kmag) - 1.0)) for rho in range(w)]
    for phi in range(h):
        cp, sp = math.cos(kangle * phi), math.sin(kangle * phi)
        for rho in range(w):
            mx[phi, rho] = np.float32(rhos[rho] * cp + center[0])
            my[phi, rho] = np.float32(rhos[rho] * sp + center[1])
    return mx, my


def remap_linear_u8(img: np.ndarray, mapx: np.ndarray, mapy: np.ndarray) -> np.ndarray:
    """cv2.remap(img, mapx, mapy, INTER_LINEAR, BORDER_CONSTANT, 0) for a uint8 HxWxC image and float32 maps: coordinates to 1/32 px
    (cvRound(map * 32), integer part saturated to short), the 15-bit bilinear table (exact for 1/32 steps), taps outside the
    image = 0, dst = (sum + 2^14) >> 15."""
    Hh, Ww, C = img.shape
    X = _cv_round(mapx.astype(np.float32) * np.float32(32.0))
    Y = _cv_round(mapy.astype(np.float32) * np.float32(32.0))
    sx, sy, ax, ay = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767), X & 31, Y & 31
    w = [(32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32]
    s = img.astype(np.int64)
    acc = np.zeros(mapx.shape + (C,), np.int64)
    for (dy, dx), wk in zip(((0, 0), (0, 1), (1, 0), (1, 1)), w):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < Hh) & (xx >= 0) & (xx < Ww)
        v = s[np.clip(yy, 0, Hh - 1), np.clip(xx, 0, Ww - 1)] * ok[..., None]
        acc += v * wk[..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def get_polar_img(img: np.ndarray, original=None) -> np.ndarray:
    """getPolarImg, hdn/models/logpolar.py:11-29: cv2.logPolar(img, o, m, WARP_FILL_OUTLIERS + INTER_LINEAR) with
    maxRadius = W / 2, m = W / log(maxRadius), o = round(original) or (H // 2, W // 2)."""
    sz = img.shape
    m = sz[1] / math.log(sz[1] / 2)
    o = tuple(np.round(original)) if original is not None else (sz[0] // 2, sz[1] // 2)
    mx, my = log_polar_maps(sz[1], sz[0], (float(o[0]), float(o[1])), m)
    return remap_linear_u8(img, mx, my)


def _inv3(m):
    a, b, c, d, e, f, g, h, i = np.asarray(m, np.float64).reshape(-1)
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    idet = 1.0 / det if det != 0.0 else 0.0
    return np.array([[(e * i - f * h), (c * h - b * i), (b * f - c * e)],
                     [(f * g - d * i), (a * i - c * g), (c * d - a * f)],
                     [(d * h - e * g), (b * g - a * h), (a * e - b * d)]], np.float64) * idet


def warp_perspective_u8(img: np.ndarray, M: np.ndarray) -> np.ndarray:
    """cv2.warpPerspective(img, M, (W, H), borderMode=BORDER_REPLICATE) for a uint8 HxWxC frame: source coordinates in
    1/32 px (blocks of 32 x 32 destination pixels for a big frame: bw = min(32*32/min(16,H), W)...), 15-bit fixed-point
    bilinear weights (exact for 1/32 steps), dst = (sum + 2^14) >> 15."""
    Hh, Ww, C = img.shape
    m = _inv3(M)
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
    Wd = np.where(Wd != 0.0, 32.0 / np.where(Wd != 0.0, Wd, 1.0), 0.0)
    X = _cv_round(np.clip((X0 + m[0, 0] * x1) * Wd, -2147483648.0, 2147483647.0))
    Y = _cv_round(np.clip((Y0 + m[1, 0] * x1) * Wd, -2147483648.0, 2147483647.0))
    sx, sy, ax, ay = X >> 5, Y >> 5, X & 31, Y & 31
    w = [(32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32]   # x 32768 / 1024
    x0, x1c = np.clip(sx, 0, Ww - 1), np.clip(sx + 1, 0, Ww - 1)
    y0, y1c = np.clip(sy, 0, Hh - 1), np.clip(sy + 1, 0, Hh - 1)
    s = img.astype(np.int64)
    acc = (s[y0, x0] * w[0][..., None] + s[y0, x1c] * w[1][..., None] + s[y1c, x0] * w[2][..., None] + s[y1c, x1c] * w[3][..., None])
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def _cubic_tab():
    """OpenCV's 32-entry bicubic coefficient table (A = -0.75, interpolateCubic) as float32."""
    A = np.float32(-0.75)
    x = (np.arange(32) / np.float32(32.0)).astype(np.float32)
    c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c1 = ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    c3 = np.float32(1.0) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], 1).astype(np.float32)   # [32, 4]


_CUBIC_ITAB = None


def cubic_itab():
    """BicubicTab_i of initInterTab2D: short(vy*vx*32768) with the row sum forced to 32768 by adjusting the largest (sum too
    small) or smallest (sum too large) of the four taps k1, k2 in {2, 3} (imgwarp.cpp's loop bounds ksize/2 .. ksize/2+1)."""
    global _CUBIC_ITAB
    if _CUBIC_ITAB is None:
        t = _cubic_tab()
        it = np.zeros((32, 32, 4, 4), np.int64)
        for i in range(32):
            for j in range(32):
                v = (t[i][:, None] * t[j][None, :]).astype(np.float32)
                q = np.clip(_cv_round(v * np.float32(32768.0)), -32768, 32767)
                diff = int(q.sum()) - 32768
                if diff != 0:
                    Mk, mk = (2, 2), (2, 2)
                    for k1 in (2, 3):
                        for k2 in (2, 3):
                            if q[k1, k2] < q[mk]:
                                mk = (k1, k2)
                            elif q[k1, k2] > q[Mk]:
                                Mk = (k1, k2)
                    if diff < 0:
                        q[Mk] -= diff
                    else:
                        q[mk] -= diff
                it[i, j] = q
        _CUBIC_ITAB = it
    return _CUBIC_ITAB


def warp_affine_cubic_u8(img: np.ndarray, M2x3: np.ndarray) -> np.ndarray:
    """cv2.warpAffine(img, M, (W, H), flags=INTER_CUBIC, borderMode=BORDER_REPLICATE) for uint8: M is inverted (float64,
    OpenCV's closed form), coordinates in 1/1024 px: adelta[x] = cvRound(M00*x*1024), X0 = cvRound((M01*y + M02)*1024) + 16,
    X = (X0 + adelta[x]) >> 5 (1/32 px), 4 x 4 taps with the 15-bit table above, dst = saturate((sum + 2^14) >> 15)."""
    Hh, Ww, C = img.shape
    M = np.asarray(M2x3, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = A11, M[0, 1] * (-D), M[1, 0] * (-D), A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    x = np.arange(Ww, dtype=np.float64)
    y = np.arange(Hh, dtype=np.float64)
    adelta, bdelta = _cv_round(M[0, 0] * x * 1024), _cv_round(M[1, 0] * x * 1024)
    X0 = _cv_round((M[0, 1] * y + M[0, 2]) * 1024) + 16
    Y0 = _cv_round((M[1, 1] * y + M[1, 2]) * 1024) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy, ax, ay = (X >> 5) - 1, (Y >> 5) - 1, X & 31, Y & 31
    tab = cubic_itab()[ay, ax]                    # [H, W, 4, 4]
    s = img.astype(np.int64)
    acc = np.zeros((Hh, Ww, C), np.int64)
    for k1 in range(4):
        yy = np.clip(sy + k1, 0, Hh - 1)
        for k2 in range(4):
            xx = np.clip(sx + k2, 0, Ww - 1)
            acc += s[yy, xx] * tab[:, :, k1, k2][..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def warp_affine_linear_f32(img: np.ndarray, M2x3: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.warpAffine(img, M, (dw, dh)) for a float32 HxW image with the default flags (INTER_LINEAR, BORDER_CONSTANT, value 0):
    the same 1/1024-px coordinate walk as warp_affine_cubic_u8, 2 x 2 taps weighted by BilinearTab_f (float32 products
    (1 - fy)(1 - fx), ... for 1/32-px fractions), taps outside the image contribute 0, summed left to right in float32."""
    img = np.asarray(img, np.float32)
    Hh, Ww = img.shape
    M = np.asarray(M2x3, np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = A11, M[0, 1] * (-D), M[1, 0] * (-D), A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    x = np.arange(dw, dtype=np.float64)
    y = np.arange(dh, dtype=np.float64)
    adelta, bdelta = _cv_round(M[0, 0] * x * 1024), _cv_round(M[1, 0] * x * 1024)
    X0 = _cv_round((M[0, 1] * y + M[0, 2]) * 1024) + 16
    Y0 = _cv_round((M[1, 1] * y + M[1, 2]) * 1024) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    one = np.float32(1.0)
    fx = (X & 31).astype(np.float32) * np.float32(1.0 / 32.0)
    fy = (Y & 31).astype(np.float32) * np.float32(1.0 / 32.0)
    w = [(one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx]
    acc = np.zeros((dh, dw), np.float32)
    for (dy, dx), wk in zip(((0, 0), (0, 1), (1, 0), (1, 1)), w):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < Hh) & (xx >= 0) & (xx < Ww)
        acc = acc + np.where(ok, img[np.clip(yy, 0, Hh - 1), np.clip(xx, 0, Ww - 1)], np.float32(0.0)) * wk
    return acc


def rot_matrix_2x3(cx, cy, rot):
    """img_rot_around_center's matrix (hdn/utils/transform.py:80-97)."""
    cc, ss = np.cos(rot), np.sin(rot)
    return np.array([[cc, -ss, cx - cx * cc + cy * ss], [ss, cc, cy - cy * cc - cx * ss]], np.float64)
