#!/usr/bin/env python3
"""Seeded synthetic planar-target sequence (SURVEY.md §8d cfg 4: POT-210 and OpenCV are not available here): a band-limited
noise texture (the target) on a smooth background, moved by a smooth random-walk homography about the frame centre.
Returns uint8 BGR frames and the ground-truth corner tracks, in the form the tracker's init / track_new take them.

    from tools.synth_sequence import make_sequence
    frames, corners, init = make_sequence(n_frames=501, frame_hw=(720, 1280), target_wh=(300, 200), seed=20260928)
    tracker.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
"""
from __future__ import annotations

import numpy as np


def _band_limited(g, h, w, sigma, channels=3):
    f = np.fft.rfft2(g.standard_normal((channels, h, w)))
    ky, kx = np.meshgrid(np.fft.fftfreq(h), np.fft.rfftfreq(w), indexing="ij")
    t = np.fft.irfft2(f * np.exp(-(kx ** 2 + ky ** 2) / (2 * sigma ** 2)), s=(h, w))
    t = (t - t.mean(axis=(1, 2), keepdims=True)) / t.std(axis=(1, 2), keepdims=True)
    return np.clip(128 + 48 * t, 0, 255).transpose(1, 2, 0)


def _bilinear(img, x, y):
    h, w, _ = img.shape
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    fx, fy = (x - x0)[..., None], (y - y0)[..., None]
    x0c, x1c, y0c, y1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1), np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    return (img[y0c, x0c] * (1 - fx) * (1 - fy) + img[y0c, x1c] * fx * (1 - fy) + img[y1c, x0c] * (1 - fx) * fy + img[y1c, x1c] * fx * fy)


def make_sequence(n_frames=501, frame_hw=(720, 1280), target_wh=(300, 200), seed=20260928, step=0.6, damping=0.97, off_decay=0.995):
    """step / damping / off_decay shape the corner random walk (defaults: the short test sequences; LONG_WALK below keeps a 501-frame
    sequence — the POT frame count, hdn/core/config.py:285 — inside the frame: corner excursions of a few tens of pixels)."""
    g = np.random.default_rng(seed)
    H, W = frame_hw
    tw, th = target_wh
    bg = _band_limited(g, H, W, 0.01)
    tex = _band_limited(g, th, tw, 0.08)
    x0, y0 = (W - tw) / 2.0, (H - th) / 2.0
    base = np.array([[x0, y0], [x0, y0 + th], [x0 + tw, y0 + th], [x0 + tw, y0]], np.float64)   # TL, BL, BR, TR (the head's order)
    bg8 = np.clip(np.rint(bg), 0, 255).astype(np.uint8)
    off = np.zeros((4, 2))
    vel = np.zeros((4, 2))
    frames, corners = [], []
    for t in range(n_frames):
        if t > 0:
            vel = damping * vel + step * g.standard_normal((4, 2))
            off = off_decay * off + vel
        dst = base + off
        # homography target (tex coords) -> frame through the 4 corners, then inverse-map every frame pixel
        src = np.array([[0, 0], [0, th], [tw, th], [tw, 0]], np.float64)
        A, b = [], []
        for (sx, sy), (dx, dy) in zip(src, dst):
            A += [[sx, sy, 1, 0, 0, 0, -dx * sx, -dx * sy], [0, 0, 0, sx, sy, 1, -dy * sx, -dy * sy]]
            b += [dx, dy]
        Hm = np.append(np.linalg.solve(np.array(A), np.array(b)), 1.0).reshape(3, 3)
        Hi = np.linalg.inv(Hm)
        # (only the bounding box of the target's quad is inverse-mapped: every pixel outside it is background)
        bx0, by0 = max(int(np.floor(dst[:, 0].min())) - 2, 0), max(int(np.floor(dst[:, 1].min())) - 2, 0)
        bx1, by1 = min(int(np.ceil(dst[:, 0].max())) + 3, W), min(int(np.ceil(dst[:, 1].max())) + 3, H)
        yy, xx = np.meshgrid(np.arange(by0, by1, dtype=np.float64), np.arange(bx0, bx1, dtype=np.float64), indexing="ij")
        den = Hi[2, 0] * xx + Hi[2, 1] * yy + Hi[2, 2]
        u, v = (Hi[0, 0] * xx + Hi[0, 1] * yy + Hi[0, 2]) / den, (Hi[1, 0] * xx + Hi[1, 1] * yy + Hi[1, 2]) / den
        inside = (u >= 0) & (u <= tw - 1) & (v >= 0) & (v <= th - 1)
        fr = np.where(inside[..., None], _bilinear(tex, np.clip(u, 0, tw - 1), np.clip(v, 0, th - 1)), bg[by0:by1, bx0:bx1])
        full = bg8.copy()
        full[by0:by1, bx0:bx1] = np.clip(np.rint(fr), 0, 255).astype(np.uint8)
        frames.append(full)
        corners.append(dst.astype(np.float32))
    c0 = corners[0]
    cx, cy = c0[:, 0].mean(), c0[:, 1].mean()
    init = {"bbox": [float(c0[:, 0].min()), float(c0[:, 1].min()), float(tw), float(th)], "poly": [float(cx), float(cy), float(tw), float(th), 0.0],
            "gt_points": c0.reshape(-1).tolist(), "first_point": c0[0].tolist()}
    return frames, corners, init


LONG_WALK = dict(step=0.35, damping=0.9, off_decay=0.97)


def success_4pts_error(pred, gt):
    """toolkit/utils/statistics.py:206-218: sqrt(sum ||delta||^2 / 4) over the 4 corners."""
    d = np.asarray(pred, np.float64).reshape(4, 2) - np.asarray(gt, np.float64).reshape(4, 2)
    return float(np.sqrt((d ** 2).sum() / 4.0))


if __name__ == "__main__":
    fr, co, init = make_sequence(n_frames=5, frame_hw=(360, 640), target_wh=(150, 100))
    print(len(fr), fr[0].shape, fr[0].dtype, co[-1], init)
