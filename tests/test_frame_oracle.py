"""CPU tests of oracle/frame_oracle.py: the pinned part against fixtures produced by the reference's own get_subwindow* /
get_search_info (tests/golden/frame.npz), the OpenCV restatements (parity unpinned) through properties."""
import numpy as np

from conftest import load_golden
from oracle import frame_oracle as F
from oracle import hdn_oracle as O


import pytest


@pytest.mark.parametrize("fixture", ["frame", "frame303"])     # frame303: model_sz = 303, the search crop of BASELINE configs[4]
def test_get_subwindow_matches_reference_crops_and_padding(fixture):
    g = load_golden(fixture)
    im, avg = g["im"], g["avg"]
    for i, (pos, sz) in enumerate(zip(g["pos"], g["sz"])):
        sz = int(sz)
        a = F.get_subwindow(im, pos, sz, sz, avg)
        assert a.dtype == np.float32 and a.shape == (1, 3, sz, sz)
        np.testing.assert_array_equal(a[0].astype(np.uint8), g[f"crop{i}"][0], err_msg=f"case {i}")
        b, pts = F.get_subwindow_for_homo(im, pos, sz, sz, avg)
        np.testing.assert_array_equal(b, a)
        np.testing.assert_array_equal(np.array(pts, np.float64), g[f"pts{i}"])
    # padding is uint8(avg): truncation, not rounding
    a = F.get_subwindow(im, [-4.0, 50.0], 25, 25, [10.7, 20.2, 30.9])
    assert tuple(a[0, :, 0, 0]) == (10.0, 20.0, 30.0)


def test_search_info_matches_reference():
    g = load_golden("frame")
    crop = F.get_subwindow(g["im"], [66.0, 48.0], 127, 127, g["avg"])
    s = F.search_info(crop[0])
    assert s.dtype == np.float64 and s.shape == (1, 127, 127)
    np.testing.assert_allclose(s, g["search_info"], rtol=0, atol=1e-12)
    # the older helper in hdn_oracle is the same arithmetic
    np.testing.assert_allclose(O.gray_normalise(crop[0].transpose(1, 2, 0)), s, rtol=0, atol=1e-6)


def test_opencv_restatements_properties():
    r = np.random.default_rng(0)
    img = r.integers(0, 256, (90, 120, 3)).astype(np.uint8)
    # identities
    assert np.array_equal(F.warp_perspective_u8(img, np.eye(3)), img)
    assert np.array_equal(F.warp_affine_cubic_u8(img, np.array([[1, 0, 0], [0, 1, 0.0]])), img)
    assert np.array_equal(F.resize_linear_u8(img, 120, 90), img)
    # integer shifts move pixels exactly and replicate the border
    w = F.warp_affine_cubic_u8(img, np.array([[1, 0, 5.0], [0, 1, -3.0]]))
    assert np.array_equal(w[:87, 5:], img[3:, :115]) and np.array_equal(w[:, :5], np.repeat(w[:, 5:6], 5, axis=1))
    w = F.warp_perspective_u8(img, np.array([[1, 0, -7.0], [0, 1, 2.0], [0, 0, 1]]))
    assert np.array_equal(w[2:, :113], img[:88, 7:]) and np.array_equal(w[:2, :113], np.repeat(img[:1, 7:], 2, axis=0))
    # half-pixel shift of the bilinear warp = rounded mean of neighbours
    w = F.warp_perspective_u8(img, np.array([[1, 0, -0.5], [0, 1, 0], [0, 0, 1.0]]))
    want = (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1
    assert np.array_equal(w[:, :-1], want)
    # every row of the fixed-point bicubic table sums to 1.0 (32768)
    it = F.cubic_itab()
    assert it.sum(axis=(2, 3)).min() == it.sum(axis=(2, 3)).max() == 32768
    # resize: constant images stay constant, 2x up-sampling of a ramp stays monotone, range never widens
    const = np.full((40, 50, 3), 137, np.uint8)
    assert (F.resize_linear_u8(const, 255, 255) == 137).all()
    ramp = np.tile(np.arange(64, dtype=np.uint8)[None, :, None] * 4, (8, 1, 3))
    up = F.resize_linear_u8(ramp, 128, 16)
    assert (np.diff(up[0, :, 0].astype(int)) >= 0).all() and up.min() == ramp.min() and up.max() == ramp.max()
    big = F.resize_linear_u8(img, 255, 255)
    assert big.min() >= img.min() and big.max() <= img.max()
    # the tracker's crop with a resize is the patch resized (shape and dtype contract of get_subwindow)
    a = F.get_subwindow(img, [60.0, 45.0], 127, 53.0, [1, 2, 3])
    assert a.shape == (1, 3, 127, 127) and a.dtype == np.float32
    np.testing.assert_array_equal(a[0].transpose(1, 2, 0), F.resize_linear_u8(F.subwindow_patch(img, [60.0, 45.0], 53.0, [1, 2, 3]), 127, 127))
    # a fractional original_sz (init_s_z_sm * scale_delta) yields floor(sz) rows, as the reference's slice does
    assert F.subwindow_patch(img, [60.0, 45.0], 53.7, [1, 2, 3]).shape == (53, 53, 3)


def test_similarity_homography_builder_matches_reference():
    """oracle.tracker_oracle.rot_scale_around_center_shift_tran against the reference's hdn/utils/transform.py:250-298 (the
    product builds H_sim on the device, hdn_similarity_logpolar_f32: tests/test_gpu_tracker.py holds it to this and to
    tests/golden/similarity.npz)."""
    from oracle.tracker_oracle import rot_scale_around_center_shift_tran
    g = load_golden("frame")
    for row, H in zip(g["sim_params"], g["sim_H"]):
        np.testing.assert_allclose(rot_scale_around_center_shift_tran(*row), H, rtol=0, atol=1e-12)


def test_log_polar_restatement_properties():
    """cv2.logPolar restated (parity-unpinned: OpenCV is in neither tree): what the algorithm must do whatever the data."""
    r = np.random.default_rng(3)
    img = r.integers(0, 256, (127, 127, 3), dtype=np.uint8)
    mx, my = F.log_polar_maps(127, 127, (63.0, 63.0), 127 / np.log(63.5))
    assert mx.dtype == my.dtype == np.float32 and mx.shape == my.shape == (127, 127)
    # rho = 0 is the centre for every angle; the last column is the circle of radius exp(126 log(63.5) / 127) - 1
    assert np.all(mx[:, 0] == 63.0) and np.all(my[:, 0] == 63.0)
    rad = np.hypot(mx[:, -1].astype(np.float64) - 63, my[:, -1].astype(np.float64) - 63)
    np.testing.assert_allclose(rad, np.exp(126 * np.log(63.5) / 127) - 1, rtol=1e-6)
    # angle runs along y: row h/4 looks straight down (+y), row 0 to the right (+x)
    assert my[0, -1] == 63.0 and mx[0, -1] > 120 and abs(float(mx[32, -1]) - 63.0) < 2.0 and my[32, -1] > 120
    lp = F.get_polar_img(img)
    assert lp.shape == img.shape and lp.dtype == np.uint8
    assert np.all(lp[:, 0] == img[63, 63])                       # integer coordinates reproduce the pixel exactly
    # a constant image stays constant wherever all four taps are inside; outliers (taps outside) pull towards the border value 0
    const = np.full((31, 31, 1), 200, np.uint8)
    mx2, my2 = F.log_polar_maps(31, 31, (15.0, 15.0), 31 / np.log(15.5))
    out = F.remap_linear_u8(const, mx2, my2)
    inside = (mx2 >= 0) & (mx2 <= 30) & (my2 >= 0) & (my2 <= 30)
    assert np.all(out[inside] == 200)
    far = F.remap_linear_u8(const, mx2 + 100, my2)
    assert np.all(far == 0)                                        # WARP_FILL_OUTLIERS: constant border 0
    # the remap itself: an integer shift map is a plain shift with zero fill
    yy, xx = np.mgrid[0:31, 0:31].astype(np.float32)
    rnd = r.integers(0, 256, (31, 31, 2), dtype=np.uint8)
    sh = F.remap_linear_u8(rnd, xx + 3, yy - 2)
    assert np.array_equal(sh[2:, :28], rnd[:29, 3:]) and np.all(sh[:2] == 0) and np.all(sh[:, 28:] == 0)
    # half-pixel map: the rounded mean of the two neighbours (weights 16384 / 16384)
    hp = F.remap_linear_u8(rnd, xx + 0.5, yy)
    np.testing.assert_array_equal(hp[:, :30], ((rnd[:, :30].astype(int) + rnd[:, 1:].astype(int)) * 16384 + 16384 >> 15).astype(np.uint8))
