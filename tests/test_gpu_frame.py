"""GPU tests (-m gpu) of the device-resident frame handling (hdn_amd.frame) through the C ABI: bit-exact against the
fixtures the reference's own get_subwindow* / get_search_info produced (tests/golden/frame.npz) for the pinned arithmetic,
and bit-exact against oracle/frame_oracle.py for the restated OpenCV pieces (parity-unpinned vs cv2 itself)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import frame_oracle as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


from hdn_amd import frame as FR  # noqa: E402


def test_get_subwindow_golden_config5_crop_size(dev):
    """hdn_subwindow_f32 at model_sz = 303 (the search crop of BASELINE configs[4]) against the reference's own get_subwindow /
    get_subwindow_for_homo (tests/golden/frame303.npz): crop position, every padding side, uint8(avg) fill — bit-exact."""
    g = load_golden("frame303")
    fr = FR.upload(g["im"])
    for i, (pos, sz) in enumerate(zip(g["pos"], g["sz"])):
        sz = int(sz)
        assert sz == 303
        a = FR.get_subwindow(fr, pos, sz, sz, g["avg"])
        assert a.shape == (1, 3, 303, 303)
        np.testing.assert_array_equal(a.cpu().numpy()[0].astype(np.uint8), g[f"crop{i}"][0], err_msg=f"case {i}")
        _, pts = FR.get_subwindow_for_homo(fr, pos, sz, sz, g["avg"])
        np.testing.assert_array_equal(np.array(pts, np.float64), g[f"pts{i}"])
    # with the resize in front (s_x != 303), against the oracle's restated cv2.resize (parity-unpinned)
    for pos, osz in (((208.0, 165.0), 380.0), ((20.3, 300.7), 251.0)):
        np.testing.assert_array_equal(FR.get_subwindow(fr, pos, 303, osz, g["avg"]).cpu().numpy(), F.get_subwindow(g["im"], pos, 303, osz, g["avg"]))


def test_get_subwindow_golden(dev):
    g = load_golden("frame")
    fr = FR.upload(g["im"])
    for i, (pos, sz) in enumerate(zip(g["pos"], g["sz"])):
        sz = int(sz)
        a = FR.get_subwindow(fr, pos, sz, sz, g["avg"])
        assert a.shape == (1, 3, sz, sz) and a.dtype == torch.float32
        np.testing.assert_array_equal(a.cpu().numpy()[0].astype(np.uint8), g[f"crop{i}"][0], err_msg=f"case {i}")
        b, pts = FR.get_subwindow_for_homo(fr, pos, sz, sz, g["avg"])
        assert torch.equal(a, b)
        np.testing.assert_array_equal(np.array(pts, np.float64), g[f"pts{i}"])
    s = FR.get_search_info(fr, [66.0, 48.0], 127, g["avg"])
    assert s.shape == (1, 1, 127, 127)
    np.testing.assert_allclose(s.cpu().numpy()[0], g["search_info"].astype(np.float32), rtol=0, atol=1e-6)
    # parameters that already live on the device
    p = torch.tensor([66.0, 48.0, 127.0, *g["avg"]], dtype=torch.float64, device=dev)
    assert torch.equal(FR.get_search_info(fr, None, None, None, params=p), s)


def test_subwindow_with_resize_vs_oracle(dev):
    r = np.random.default_rng(5)
    im = r.integers(0, 256, (360, 640, 3)).astype(np.uint8)
    fr = FR.upload(im)
    avg = np.mean(im, axis=(0, 1))
    for pos, osz, msz in (((320.0, 180.0), 253.0, 255), ((10.5, 350.2), 311.0, 255), ((600.0, 20.0), 95.0, 127), ((300.3, 200.7), 57.6, 127),
                          ((100.0, 100.0), 127.0, 127), ((333.0, 111.0), 510.0, 255), ((5.0, 5.0), 30.0, 127)):
        want = F.get_subwindow(im, pos, msz, osz, avg)
        got = FR.get_subwindow(fr, pos, msz, osz, avg)
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=str((pos, osz, msz)))
        gray = FR.get_search_info(fr, pos, osz, avg, model_sz=msz)
        np.testing.assert_allclose(gray.cpu().numpy()[0], F.search_info(want[0]).astype(np.float32), rtol=0, atol=1e-6)


def test_frame_warps_vs_oracle(dev):
    r = np.random.default_rng(6)
    im = r.integers(0, 256, (180, 320, 3)).astype(np.uint8)
    fr = FR.upload(im)
    assert torch.equal(FR.warp_perspective(fr, np.eye(3)), fr)
    assert torch.equal(FR.rot_around_center(fr, 100.0, 80.0, 0.0), fr)
    Hs = [np.array([[1.02, 0.03, -4.2], [-0.02, 0.97, 6.1], [1e-5, -2e-5, 1.0]]),
          np.array([[0.9, 0.2, 30.0], [-0.15, 1.1, -12.0], [3e-4, 1e-4, 1.0]]),
          np.array([[1, 0, 500.0], [0, 1, -400.0], [0, 0, 1.0]])]
    for M in Hs:
        got = FR.warp_perspective(fr, M).cpu().numpy()
        np.testing.assert_array_equal(got, F.warp_perspective_u8(im, M))
    Md = torch.tensor(Hs[0].reshape(-1), dtype=torch.float64, device=dev)
    assert torch.equal(FR.warp_perspective(fr, Md), FR.warp_perspective(fr, Hs[0]))
    for (cx, cy, rot) in ((160.0, 90.0, 0.3), (10.0, 170.0, -1.2), (400.0, -20.0, 3.0)):
        got = FR.rot_around_center(fr, cx, cy, rot).cpu().numpy()
        np.testing.assert_array_equal(got, F.warp_affine_cubic_u8(im, F.rot_matrix_2x3(cx, cy, rot)))
    # sizes whose pixel count is not a multiple of 4 (the last pixels leave as bytes, the rest as packed 4-byte stores), images narrower
    # than the 4-tap window, a single pixel; and 1- / 4-channel frames (byte path throughout)
    for (h, w) in ((7, 5), (33, 31), (1, 1), (2, 3), (5, 127), (257, 3)):
        im2 = r.integers(0, 256, (h, w, 3)).astype(np.uint8)
        f2 = FR.upload(im2)
        M = np.array([[0.97, 0.05, 0.7], [-0.04, 1.03, -0.4], [1e-4, -2e-4, 1.0]])
        np.testing.assert_array_equal(FR.warp_perspective(f2, M).cpu().numpy(), F.warp_perspective_u8(im2, M), err_msg=str((h, w)))
        A = F.rot_matrix_2x3(w / 2.0, h / 3.0, 0.4)
        np.testing.assert_array_equal(FR.warp_affine_cubic(f2, A).cpu().numpy(), F.warp_affine_cubic_u8(im2, A), err_msg=str((h, w)))
    for c in (1, 4):
        im2 = r.integers(0, 256, (21, 19, c)).astype(np.uint8)
        f2 = FR.upload(im2)
        M = np.array([[1.01, 0.02, -1.3], [0.03, 0.98, 2.2], [0, 0, 1.0]])
        np.testing.assert_array_equal(FR.warp_perspective(f2, M).cpu().numpy(), F.warp_perspective_u8(im2, M))
        A = F.rot_matrix_2x3(9.0, 11.0, -0.2)
        np.testing.assert_array_equal(FR.warp_affine_cubic(f2, A).cpu().numpy(), F.warp_affine_cubic_u8(im2, A))
    with pytest.raises(Exception):
        FR.get_subwindow(torch.zeros(4, 4, 3, dtype=torch.uint8), [1, 1], 3, 3, [0, 0, 0])


def test_log_polar_vs_oracle(dev):
    """get_polar_img / get_subwindow(islog=1) (restated cv2.logPolar, parity-unpinned) bit-exact against the oracle's restatement."""
    r = np.random.default_rng(11)
    for S, C in ((127, 3), (31, 1), (64, 2)):
        img = r.integers(0, 256, (S, S, C), dtype=np.uint8)
        patch = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))[None].astype(np.float32)).to(dev)
        got = FR.get_polar_img(patch)
        assert got.shape == patch.shape and got.dtype == torch.float32
        np.testing.assert_array_equal(got.cpu().numpy()[0].transpose(1, 2, 0).astype(np.uint8), F.get_polar_img(img), err_msg=f"S={S}")
        assert torch.equal(got, got.round())                                       # uint8-valued
    o = (40.2, 71.6)
    patch = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))[None].astype(np.float32)).to(dev)
    np.testing.assert_array_equal(FR.get_polar_img(patch, original=o).cpu().numpy()[0].transpose(1, 2, 0).astype(np.uint8),
                                  F.get_polar_img(img, original=o))
    # the 6-channel template crop of the tracker: np.concatenate((im_patch, getPolarImg(im_patch)), 2)
    g = load_golden("frame")
    fr = FR.upload(g["im"])
    six = FR.get_subwindow(fr, g["pos"][0], 127, int(g["sz"][0]), g["avg"], islog=1)
    assert six.shape == (1, 6, 127, 127)
    three = FR.get_subwindow(fr, g["pos"][0], 127, int(g["sz"][0]), g["avg"])
    assert torch.equal(six[:, :3], three)
    ref = F.get_polar_img(three.cpu().numpy()[0].transpose(1, 2, 0).astype(np.uint8))
    np.testing.assert_array_equal(six[0, 3:].cpu().numpy().transpose(1, 2, 0).astype(np.uint8), ref)
    with pytest.raises(ValueError):
        FR.get_polar_img(patch[0])
