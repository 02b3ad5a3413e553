"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the
golden vectors captured from the reference.

Tolerances (fp32 path, BASELINE.json north_star: 1e-4 abs on the predicted corner displacements):
  * correlations: a tap sum of n products is only defined to fp32 summation order.  Bound used:
        |hip - ref| <= 1e-4 + 2e-6 * sum|x*k|        (abs 1e-4 for the O(1..10) production outputs)
    and, against the float64 truth, the HIP error may not exceed twice the reference's own worst error.
  * PreShareFeature / warp / scores: 1e-4 abs (observed ~1e-6).
  * DLT H_mat: 1e-5 abs for the production corners; the fp32 reference is itself ~1.5e-4 off the exact
    solution for general quadrilaterals (tests/test_oracle_golden.py::test_dlt_solve), so those compare to
    the float64 solution at 1e-5 and to the reference at 5e-4.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, golden_rng, load_golden, relu_normal
from oracle import hdn_oracle as O

pytestmark = pytest.mark.gpu

torch.set_num_threads(max(1, torch.get_num_threads()))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


import hdn_amd  # noqa: E402
from hdn_amd import homography as G  # noqa: E402
from hdn_amd import xcorr as X  # noqa: E402


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def cases(npz, suffix="__x"):
    return sorted(k[: -len(suffix)] for k in npz.files if k.endswith(suffix))


def check_xcorr(got, x, k, ref, circular, name):
    truth = (O.xcorr_depthwise_circular_f64 if circular else O.xcorr_depthwise_f64)(x, k)
    mag = (O.xcorr_depthwise_circular_f64 if circular else O.xcorr_depthwise_f64)(np.abs(x), np.abs(k))
    got = got.cpu().numpy()
    assert got.shape == ref.shape, name
    assert np.all(np.abs(got - ref) <= 1e-4 + 2e-6 * mag), f"{name}: max|hip-ref|={np.abs(got - ref).max():.3e}"
    e_hip, e_ref = np.abs(got - truth).max(), np.abs(ref - truth).max()
    assert e_hip <= 2 * e_ref + 1e-6, f"{name}: hip err {e_hip:.3e} vs reference err {e_ref:.3e} against float64"


# --------------------------------------------------------------------------- correlations
def test_xcorr_depthwise_golden(dev):
    g = load_golden("xcorr_depthwise")
    seen = set()
    for variant in ("fft", "direct"):  # only the 31x31 (x) 61x61 fixtures depend on it
        with X.north_variant(variant):
            for n in cases(g):
                x, k = g[n + "__x"], g[n + "__k"]
                if variant != "fft" and x.shape[-1] != 61:
                    continue
                y = hdn_amd.xcorr_depthwise(T(x).to(dev), T(k).to(dev))
                seen.add(X.last_variant())
                check_xcorr(y, x, k, g[n + "__y"], False, n + "/" + variant)
    assert X.current_north_variant() == "fft"
    # the fixtures exercise the specialised kernels (both families for the north-star shape) and the generic one
    assert {"prod_29x29_5x5", "cfg5_35x35_5x5", "north_61x61_31x31", "north_fftc_61x61_31x31", "generic_lds"} <= seen, seen


def test_xcorr_depthwise_sampled_full_channel(dev):
    g = load_golden("xcorr_depthwise_sampled")
    for j, n in enumerate(["prod256_5x29", "north256_31x61"]):
        B, C, Hx, Wx, Hk, Wk = (int(v) for v in g[n + "__shape"])
        r = golden_rng(150 + j)
        x, k = relu_normal(r, (B, C, Hx, Wx)), relu_normal(r, (B, C, Hk, Wk))
        y = hdn_amd.xcorr_depthwise(T(x).to(dev), T(k).to(dev)).cpu().numpy()
        val = y.reshape(-1)[g[n + "__idx"]]
        ref = g[n + "__val"]
        assert np.all(np.abs(val - ref) <= 1e-4 + 1e-5 * np.abs(ref)), (n, np.abs(val - ref).max())
        s = y.astype(np.float64).sum()
        assert abs(s - float(g[n + "__sum"])) <= 1e-6 * abs(float(g[n + "__sum"])), n


def test_xcorr_depthwise_circular_golden(dev):
    g = load_golden("xcorr_depthwise_circular")
    seen = set()
    for n in cases(g):
        x, k = g[n + "__x"], g[n + "__k"]
        y = hdn_amd.xcorr_depthwise_circular(T(x).to(dev), T(k).to(dev))
        seen.add(X.last_variant())
        check_xcorr(y, x, k, g[n + "__y"], True, n)
    assert {"circ13", "generic_lds"} <= seen, seen


@pytest.mark.parametrize("shape", [(5, 7, 29, 29, 5, 5), (3, 5, 61, 61, 31, 31), (2, 9, 35, 35, 5, 5), (1, 1, 29, 29, 5, 5)])
def test_xcorr_ragged_plane_counts_vs_oracle(dev, shape):
    """Plane counts that are not a multiple of the planes-per-workgroup group (tail workgroups, unaligned tails)."""
    B, C, Hx, Wx, Hk, Wk = shape
    r = np.random.default_rng(sum(shape))
    x, k = relu_normal(r, (B, C, Hx, Wx)), relu_normal(r, (B, C, Hk, Wk))
    y = hdn_amd.xcorr_depthwise(T(x).to(dev), T(k).to(dev))
    check_xcorr(y, x, k, O.xcorr_depthwise(T(x), T(k)).numpy(), False, str(shape))


NORTH_NAME = {"fft": "north_fftc_61x61_31x31", "direct": "north_61x61_31x31"}


@pytest.mark.parametrize("planes", [(1, 1), (1, 2), (1, 3), (1, 4), (3, 5), (2, 8), (1, 33), (5, 205), (8, 256)])
@pytest.mark.parametrize("variant", ["fft", "direct"])
def test_xcorr_north_plane_counts(dev, planes, variant):
    """31x31 (x) 61x61 for odd / even / tiny plane counts (the FFT kernel works on PAIRS of planes, its last pair(s)
    take a guarded path in an extra workgroup) up to several persistent passes; signed and post-ReLU data."""
    B, C = planes
    r = np.random.default_rng(1000 * B + C)
    for signed in (False, True):
        x = r.standard_normal((B, C, 61, 61), dtype=np.float32)
        k = r.standard_normal((B, C, 31, 31), dtype=np.float32)
        if not signed:
            x, k = np.maximum(x, 0), np.maximum(k, 0)
        with X.north_variant(variant):
            y = hdn_amd.xcorr_depthwise(T(x).to(dev), T(k).to(dev))
            assert X.last_variant() == NORTH_NAME[variant]
            assert torch.equal(y, hdn_amd.xcorr_depthwise(T(x).to(dev), T(k).to(dev)))  # deterministic
        check_xcorr(y, x, k, O.xcorr_depthwise(T(x), T(k)).numpy(), False, f"{planes} {variant} signed={signed}")


def test_xcorr_north_multi_problem_launch(dev):
    """Several 31x31 (x) 61x61 problems through the multi entry point (the FFT kernel runs them back to back)."""
    r = np.random.default_rng(404)
    xs = [T(relu_normal(r, (2, 9, 61, 61))).to(dev) for _ in range(3)]
    ks = [T(relu_normal(r, (2, 9, 31, 31))).to(dev) for _ in range(3)]
    for variant in ("fft", "direct"):
        with X.north_variant(variant):
            outs = hdn_amd.xcorr_depthwise_multi(xs, ks)
            for x, k, o in zip(xs, ks, outs):
                assert torch.equal(o, hdn_amd.xcorr_depthwise(x, k))
                check_xcorr(o, x.cpu().numpy(), k.cpu().numpy(), O.xcorr_depthwise(x.cpu(), k.cpu()).numpy(), False, variant)


def test_xcorr_north_fft_pair_crosstalk_is_rounding_only(dev):
    """The FFT kernel packs planes (2p, 2p+1) into one complex transform.  A plane next to a 1000x larger one must
    still meet the bound relative to the PAIR's magnitude, and an all-zero kernel plane gives |out| at rounding level
    of the partner (the direct kernels give exact zeros: test_xcorr_multi_eight_problems_and_limits)."""
    r = np.random.default_rng(77)
    x, k = relu_normal(r, (1, 6, 61, 61)), relu_normal(r, (1, 6, 31, 31))
    x[0, 1] *= 1000.0
    k[0, 2] = 0
    truth = O.xcorr_depthwise_f64(x, k)
    mag = O.xcorr_depthwise_f64(np.abs(x), np.abs(k))
    pair_mag = np.maximum(mag[:, 0::2], mag[:, 1::2]).repeat(2, axis=1)  # per pair of planes
    for variant in ("fft",):
        with X.north_variant(variant):
            y = hdn_amd.xcorr_depthwise(T(x).to(dev), T(k).to(dev)).cpu().numpy()
        assert np.all(np.abs(y - truth) <= 1e-4 + 2e-6 * pair_mag.max(axis=(2, 3), keepdims=True))
        assert np.abs(y[0, 2]).max() <= 2e-6 * mag[0, 3].max()
        assert np.abs(y[0, 0] - truth[0, 0]).max() <= 2e-6 * mag[0, 1].max()  # small plane beside the large one


@pytest.mark.parametrize("shape", [(3, 7, 13, 13, 13, 13), (1, 1, 13, 13, 13, 13), (2, 3, 9, 12, 4, 6), (1, 2, 5, 5, 9, 9)])
def test_xcorr_circular_ragged_vs_oracle(dev, shape):
    B, C, Hx, Wx, Hk, Wk = shape
    r = np.random.default_rng(sum(shape) + 1)
    x = r.standard_normal((B, C, Hx, Wx), dtype=np.float32)
    k = r.standard_normal((B, C, Hk, Wk), dtype=np.float32)
    y = hdn_amd.xcorr_depthwise_circular(T(x).to(dev), T(k).to(dev))
    check_xcorr(y, x, k, O.xcorr_depthwise_circular(T(x), T(k)).numpy(), True, str(shape))


def test_xcorr_generic_large_plane_uses_l2_path(dev):
    r = np.random.default_rng(5)
    x, k = relu_normal(r, (1, 2, 150, 140)), relu_normal(r, (1, 2, 3, 2))
    y = hdn_amd.xcorr_depthwise(T(x).to(dev), T(k).to(dev))
    assert X.last_variant() == "generic_l2"
    check_xcorr(y, x, k, O.xcorr_depthwise(T(x), T(k)).numpy(), False, "large plane")


def test_xcorr_unaligned_and_strided_inputs(dev):
    """A sliced tensor gives a base pointer that is only 4-byte aligned and a non-contiguous view."""
    r = np.random.default_rng(11)
    xb, kb = relu_normal(r, (3, 6, 29, 29)), relu_normal(r, (3, 6, 5, 5))
    xd, kd = T(xb).to(dev), T(kb).to(dev)
    flat = torch.zeros(xd.numel() + 1, device=dev)
    flat[1:] = xd.reshape(-1)
    x_off = flat[1:].view_as(xd)  # data_ptr % 16 == 4
    assert x_off.data_ptr() % 16 != 0
    ref = O.xcorr_depthwise(T(xb), T(kb)).numpy()
    check_xcorr(hdn_amd.xcorr_depthwise(x_off, kd), xb, kb, ref, False, "unaligned")
    x_nc = xd.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)  # same values, non-contiguous strides
    assert not x_nc.is_contiguous()
    check_xcorr(hdn_amd.xcorr_depthwise(x_nc, kd), xb, kb, ref, False, "strided")
    assert torch.equal(xd.cpu(), T(xb)) and torch.equal(kd.cpu(), T(kb))  # inputs not mutated


@pytest.mark.parametrize("shape,circ", [((3, 7, 61, 61, 31, 31), False), ((5, 3, 13, 13, 13, 13), True), ((2, 9, 35, 35, 5, 5), False)])
def test_xcorr_specialised_kernels_on_unaligned_pointers(dev, shape, circ):
    """4-byte-aligned (not 16-byte) x / k / out base pointers and ragged plane counts through every specialised kernel:
    the all-in-flight 16-byte staging paths must fall back correctly."""
    B, C, Hx, Wx, Hk, Wk = shape
    r = np.random.default_rng(sum(shape) + 7)
    xb = relu_normal(r, (B, C, Hx, Wx)) if not circ else r.standard_normal((B, C, Hx, Wx), dtype=np.float32)
    kb = relu_normal(r, (B, C, Hk, Wk)) if not circ else r.standard_normal((B, C, Hk, Wk), dtype=np.float32)

    def shifted(a, off):
        flat = torch.zeros(a.size + off, device=dev)
        flat[off:] = T(a).to(dev).reshape(-1)
        v = flat[off:].view(*a.shape)
        assert v.data_ptr() % 16 == (4 * off) % 16
        return v

    fn = hdn_amd.xcorr_depthwise_circular if circ else hdn_amd.xcorr_depthwise
    ofn = O.xcorr_depthwise_circular if circ else O.xcorr_depthwise
    ref = ofn(T(xb), T(kb)).numpy()
    for ox, ok in ((1, 0), (0, 3), (2, 1)):
        y = fn(shifted(xb, ox), shifted(kb, ok))
        check_xcorr(y, xb, kb, ref, circ, f"{shape} offsets {ox},{ok}")
        if Hx == 61:  # the column-first FFT kernel has no alignment requirement: it serves these pointers itself
            assert X.last_variant() == "north_fftc_61x61_31x31"


def test_xcorr_multi_eight_problems_and_limits(dev):
    r = np.random.default_rng(88)
    xs = [T(relu_normal(r, (1, 6, 29, 29))).to(dev) for _ in range(8)]
    ks = [T(relu_normal(r, (1, 6, 5, 5))).to(dev) for _ in range(8)]
    outs = hdn_amd.xcorr_depthwise_multi(xs, ks)
    for x, k, o in zip(xs, ks, outs):
        assert torch.equal(o, hdn_amd.xcorr_depthwise(x, k))
    with pytest.raises(ValueError):
        hdn_amd.xcorr_depthwise_multi(xs + xs[:1], ks + ks[:1])
    with pytest.raises(ValueError):
        hdn_amd.xcorr_depthwise_multi(xs[:2], [ks[0], ks[1][:, :, :3, :3]])
    # zero taps are skipped by the 31x31 kernel: an all-zero kernel plane gives exact zeros, and a signed-zero tap too
    x = T(relu_normal(r, (1, 4, 61, 61))).to(dev)
    k = T(relu_normal(r, (1, 4, 31, 31))).to(dev)
    k[0, 1] = 0
    k[0, 2, 3, 4] = -0.0
    with X.north_variant("direct"):
        y = hdn_amd.xcorr_depthwise(x, k)
    assert X.last_variant() == "north_61x61_31x31"
    assert bool((y[0, 1] == 0).all())
    check_xcorr(y, x.cpu().numpy(), k.cpu().numpy(), O.xcorr_depthwise(x.cpu(), k.cpu()).numpy(), False, "zero taps")


def test_xcorr_fast_and_slow_golden(dev):
    """The channel-contracting variants of hdn/core/xcorr.py (unselected UPChannelBAN head)."""
    g = load_golden("xcorr_fast")
    for n in ("cls_o2", "loc_o4"):
        x, k = g[n + "__x"], g[n + "__k"]
        y = hdn_amd.xcorr_fast(T(x).to(dev), T(k).to(dev)).cpu().numpy()
        assert y.shape == g[n + "__y"].shape
        mag = O.xcorr_fast(T(np.abs(x)), T(np.abs(k))).numpy()
        assert np.all(np.abs(y - g[n + "__y"]) <= 1e-4 + 2e-6 * mag), n
    ys = hdn_amd.xcorr_slow(T(g["slow__x"]).to(dev), T(g["slow__k"]).to(dev)).cpu().numpy()
    np.testing.assert_allclose(ys, g["slow__y"], rtol=0, atol=1e-4)
    with pytest.raises(ValueError):
        hdn_amd.xcorr_slow(T(g["cls_o2__x"]).to(dev), T(g["cls_o2__k"]).to(dev))  # O = 2: the reference raises too
    with pytest.raises(ValueError):
        hdn_amd.xcorr_fast(torch.zeros(1, 3, 5, 5, device=dev), torch.zeros(1, 4, 3, 3, device=dev))


@pytest.mark.parametrize("O_", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("ksz", [(5, 5), (3, 4)])
def test_xcorr_fast_every_output_width(dev, O_, ksz):
    """Every compiled instantiation of xcorr_fast_kernel<O, K> (O = 1..8 output maps; 5x5 taps unrolled, other sizes generic)
    against the oracle's F.conv2d form (hdn/core/xcorr.py:22-34), ragged channel count, signed data."""
    r = np.random.default_rng(100 * O_ + ksz[1])
    B, C = 3, 37
    x = r.standard_normal((B, C, 13, 11), dtype=np.float32)
    k = r.standard_normal((B, O_ * C, ksz[0], ksz[1]), dtype=np.float32)
    y = hdn_amd.xcorr_fast(T(x).to(dev), T(k).to(dev))
    assert torch.equal(y, hdn_amd.xcorr_fast(T(x).to(dev), T(k).to(dev)))   # fixed reduction order
    ref = O.xcorr_fast(T(x), T(k)).numpy()
    mag = O.xcorr_fast(T(np.abs(x)), T(np.abs(k))).numpy()
    assert y.shape == ref.shape
    assert np.all(np.abs(y.cpu().numpy() - ref) <= 1e-4 + 2e-6 * mag)
    if O_ == 8:
        with pytest.raises(ValueError):   # HDN_E_LIMIT: XF_MAX_O
            hdn_amd.xcorr_fast(T(x).to(dev), torch.zeros((B, 9 * C, 5, 5), device=dev))


def test_xcorr_multi_launch_equals_single(dev):
    r = np.random.default_rng(21)
    xs = [T(relu_normal(r, (2, 32, 29, 29))).to(dev) for _ in range(6)]
    ks = [T(relu_normal(r, (2, 32, 5, 5))).to(dev) for _ in range(6)]
    outs = hdn_amd.xcorr_depthwise_multi(xs, ks)
    for x, k, o in zip(xs, ks, outs):
        assert torch.equal(o, hdn_amd.xcorr_depthwise(x, k))
    xs = [T(r.standard_normal((2, 32, 13, 13), dtype=np.float32)).to(dev) for _ in range(6)]
    ks = [T(r.standard_normal((2, 32, 13, 13), dtype=np.float32)).to(dev) for _ in range(6)]
    outs = hdn_amd.xcorr_depthwise_multi(xs, ks, circular=True)
    for x, k, o in zip(xs, ks, outs):
        assert torch.equal(o, hdn_amd.xcorr_depthwise_circular(x, k))


def _full_size_properties(dev, fn, ofn, shape_x, shape_k, signed, paired=False):
    """BASELINE full sizes (B=64, C=256): size-independent properties + oracle on a few sampled planes."""
    gen = torch.Generator(device="cpu").manual_seed(20260928)
    mk = (lambda s: torch.randn(s, generator=gen)) if signed else (lambda s: torch.randn(s, generator=gen).clamp_min(0))
    x, k, k2 = mk(shape_x), mk(shape_k), mk(shape_k)
    xd, kd, k2d = x.to(dev), k.to(dev), k2.to(dev)
    y = fn(xd, kd)
    # (1) determinism / idempotence
    assert torch.equal(y, fn(xd, kd))
    # (2) linearity in the kernel: corr(x, k + 2*k2) == corr(x,k) + 2*corr(x,k2)
    lhs = fn(xd, kd + 2 * k2d)
    rhs = y + 2 * fn(xd, k2d)
    mag = fn(xd.abs(), (kd.abs() + 2 * k2d.abs()))
    assert bool(((lhs - rhs).abs() <= 1e-4 + 4e-6 * mag).all())
    # (3) checksum of checksums: sum_ij out[p] == sum_uv k[p,u,v] * S[p,u,v], S = window sums of x (float64 on CPU)
    ysum = y.double().sum(dim=(2, 3)).cpu()
    planes = [(0, 0), (shape_x[0] - 1, shape_x[1] - 1), (shape_x[0] // 2, 7), (3, shape_x[1] // 2)]
    for (b, c) in planes:
        ref = ofn(x[b:b + 1, c:c + 1], k[b:b + 1, c:c + 1])
        got = y[b, c].cpu()
        tol = 1e-4 + 1e-5 * ref.abs().max().item()
        assert (got - ref[0, 0]).abs().max().item() <= tol, (b, c)
        assert abs(ysum[b, c].item() - ref.double().sum().item()) <= 1e-5 * abs(ref.double().sum().item()) + 1e-3
    # (4) every plane is finite and planes are not mixed up: zero one plane's kernel -> only that plane is zero
    kz = kd.clone()
    kz[5, 17] = 0
    yz = fn(xd, kz)
    if not paired:
        assert bool((yz[5, 17] == 0).all())
        yz[5, 17] = y[5, 17]
        assert torch.equal(yz, y)
    else:  # planes (5,16) and (5,17) share one complex transform: they move by rounding, everything else not at all
        assert float(yz[5, 17].abs().max()) <= 2e-6 * float(mag[5, 16].max())
        assert float((yz[5, 16] - y[5, 16]).abs().max()) <= 1e-4 + 2e-6 * float(mag[5, 16].max())
        yz[5, 16:18] = y[5, 16:18]
        assert torch.equal(yz, y)


def test_xcorr_full_size_production(dev):
    _full_size_properties(dev, hdn_amd.xcorr_depthwise, O.xcorr_depthwise, (64, 256, 29, 29), (64, 256, 5, 5), False)
    assert X.last_variant() == "prod_29x29_5x5"


def test_xcorr_full_size_config5(dev):
    """BASELINE configs[4]: 303-px search window => 5x5 (x) 35x35 -> 31x31 at batch 256 (1.1 GB of search features), through
    the size-independent properties; plus ragged plane counts (tail workgroups, unaligned groups) against the oracle."""
    _full_size_properties(dev, hdn_amd.xcorr_depthwise, O.xcorr_depthwise, (256, 256, 35, 35), (256, 256, 5, 5), False)
    assert X.last_variant() == "cfg5_35x35_5x5"
    r = np.random.default_rng(35)
    for B, C in ((1, 1), (1, 3), (3, 5), (2, 9), (5, 51)):
        x, k = r.standard_normal((B, C, 35, 35), dtype=np.float32), r.standard_normal((B, C, 5, 5), dtype=np.float32)
        y = hdn_amd.xcorr_depthwise(T(x).to(dev), T(k).to(dev))
        check_xcorr(y, x, k, O.xcorr_depthwise(T(x), T(k)).numpy(), False, f"cfg5 {B}x{C}")
        # unaligned plane groups (a view that starts 4 bytes into the allocation)
        buf = torch.zeros(x.size + 1, device=dev)
        buf[1:] = T(x).to(dev).reshape(-1)
        y2 = hdn_amd.xcorr_depthwise(buf[1:].reshape(x.shape), T(k).to(dev))
        assert torch.equal(y2, y)


@pytest.mark.parametrize("variant", ["fft", "direct"])
def test_xcorr_full_size_north_star(dev, variant):
    with X.north_variant(variant):
        _full_size_properties(dev, hdn_amd.xcorr_depthwise, O.xcorr_depthwise, (64, 256, 61, 61), (64, 256, 31, 31), False,
                              paired=(variant != "direct"))
        assert X.last_variant() == NORTH_NAME[variant]


def test_xcorr_north_variant_switch_rejects_retired_values(dev):
    """hdn_xcorr_north_variant takes the two kernels that are left (ABI 5); the retired ids of ABI <= 4 come back as HDN_E_LIMIT, and
    HDN_NORTH=direct in the environment selects the direct kernel for a whole process."""
    import os
    import subprocess
    import sys
    from hdn_amd import _lib
    lib = _lib.load()
    prev = lib.hdn_xcorr_north_variant(-1)
    for v in (0, 2, 3, 4, 6):
        assert lib.hdn_xcorr_north_variant(v) == -3
    assert lib.hdn_xcorr_north_variant(-1) == prev
    with pytest.raises(ValueError):
        X.north_variant("mfma")
    code = ("import sys; sys.path.insert(0, %r); import torch, hdn_amd; from hdn_amd import xcorr as X; "
            "y = hdn_amd.xcorr_depthwise(torch.rand(1, 4, 61, 61).cuda(), torch.rand(1, 4, 31, 31).cuda()); print(X.last_variant())") % ROOT
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, HDN_NORTH="direct"))
    assert res.stdout.strip().endswith("north_61x61_31x31"), res.stdout[-500:] + res.stderr[-1500:]


def test_xcorr_full_size_circular(dev):
    _full_size_properties(dev, hdn_amd.xcorr_depthwise_circular, O.xcorr_depthwise_circular, (64, 256, 13, 13), (64, 256, 13, 13), True)
    assert X.last_variant() == "circ13"


def test_xcorr_circular_wraps_rows_and_clamps_columns(dev):
    """Rolling x by one row rolls the output by one row (circular axis); the column axis does not wrap."""
    r = np.random.default_rng(3)
    x = T(r.standard_normal((1, 4, 13, 13), dtype=np.float32)).to(dev)
    k = T(r.standard_normal((1, 4, 13, 13), dtype=np.float32)).to(dev)
    y = hdn_amd.xcorr_depthwise_circular(x, k)
    y_roll = hdn_amd.xcorr_depthwise_circular(torch.roll(x, 1, dims=2), k)
    assert torch.allclose(torch.roll(y, 1, dims=2), y_roll, atol=1e-5)
    y_rollc = hdn_amd.xcorr_depthwise_circular(torch.roll(x, 1, dims=3), k)
    assert not torch.allclose(torch.roll(y, 1, dims=3), y_rollc, atol=1e-3)


# --------------------------------------------------------------------------- PreShareFeature
def share_sd(npz, prefix):
    return {k[len(prefix):].replace("__", "."): torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}


def test_share_feature_golden(dev):
    g = load_golden("share_feature")
    m = hdn_amd.PreShareFeature()
    m.load_state_dict(share_sd(g, "sd__"))
    m = m.to(dev).eval()
    for xin, yout in (("x", "y"), ("x_small", "y_small")):
        y = m(T(g[xin]).to(dev)).cpu().numpy()
        assert y.shape == g[yout].shape
        np.testing.assert_allclose(y, g[yout], rtol=0, atol=1e-4)
        assert np.abs(y - g[yout]).max() < 2e-5  # observed: a few ulp
    # parameters changed in place -> folded block is rebuilt (version-counter tracked) ...
    y0 = m(T(g["x_small"]).to(dev))
    with torch.no_grad():
        m.ShareFeature[7].bias.add_(0.5)
    y1 = m(T(g["x_small"]).to(dev))
    assert float((y1 - y0).abs().max()) > 0.1
    # ... while writes through .data bypass the version counter and need an explicit refresh
    m.ShareFeature[7].bias.data.add_(0.5)
    m.refresh()
    y2 = m(T(g["x_small"]).to(dev))
    assert float((y2 - y1).abs().max()) > 0.1


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (3, 1, 5, 131), (2, 1, 130, 7), (64, 1, 127, 127), (2, 1, 255, 255)])
def test_share_feature_shapes_vs_oracle(dev, shape):
    torch.manual_seed(sum(shape))
    m = hdn_amd.PreShareFeature().eval()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.5, 0.5)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.3, 0.3)
    x = torch.randn(shape)
    sd = {"ShareFeature." + k: v.clone() for k, v in m.ShareFeature.state_dict().items()}
    ref = O.share_feature(x, sd)
    y = m.to(dev)(x.to(dev)).cpu()
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) <= 1e-4 + 1e-5 * float(ref.abs().max())


# --------------------------------------------------------------------------- DLT / warp
def test_dlt_solve_golden(dev):
    g = load_golden("dlt_solve")
    H = hdn_amd.DLT_solve(T(g["src"]).to(dev), T(g["off"]).to(dev)).cpu().numpy()
    assert H.shape == g["H"].shape == (64, 1, 3, 3)
    np.testing.assert_allclose(H, g["H"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(H.reshape(-1, 3, 3), O.dlt_solve_f64(g["src"], g["off"]), rtol=0, atol=2e-6)
    np.testing.assert_allclose(H[0, 0], np.eye(3), atol=1e-6)
    H2 = hdn_amd.DLT_solve(T(g["src2"]).to(dev), T(g["off2"]).to(dev)).cpu().numpy()
    np.testing.assert_allclose(H2.reshape(-1, 3, 3), O.dlt_solve_f64(g["src2"], g["off2"]), rtol=0, atol=1e-5)
    np.testing.assert_allclose(H2, g["H2"], rtol=0, atol=5e-4)


def test_dlt_solve_batch_sizes_and_point_order(dev):
    r = np.random.default_rng(9)
    for B in (1, 7, 33, 1000):
        src = np.tile(np.array([0, 0, 0, 127, 127, 127, 127, 0], np.float32), (B, 1))
        off = (8 * r.standard_normal((B, 8))).astype(np.float32)
        H = hdn_amd.DLT_solve(T(src).to(dev), T(off).to(dev)).cpu().numpy().reshape(B, 3, 3)
        np.testing.assert_allclose(H, O.dlt_solve_f64(src, off), rtol=0, atol=5e-6)
        # H maps every source corner onto its displaced position
        pts = np.concatenate([src.reshape(B, 4, 2), np.ones((B, 4, 1), np.float32)], axis=2).astype(np.float64)
        q = np.einsum("bij,bkj->bki", H.astype(np.float64), pts)
        np.testing.assert_allclose(q[..., :2] / q[..., 2:], (src + off).reshape(B, 4, 2), atol=2e-3)


def test_fused_solve_is_bit_identical_to_the_standalone_solve(dev):
    """hdn_dlt_warp_f32 solves the 8x8 system with the matrix in LDS (one wave sharing a pivot step), hdn_dlt_solve_f32 with
    wave shuffles on 8 lanes: same operations on the same values in the same order => the same bits, also for ill-conditioned
    quads and huge offsets."""
    gen = torch.Generator().manual_seed(5)
    for scale in (0.0, 1.0, 8.0, 30.0, 100.0):
        B = 256
        src = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(B, 1) + (scale / 4) * torch.randn(B, 8, generator=gen)
        off = scale * torch.randn(B, 8, generator=gen)
        img = torch.randn(B, 1, 31, 33, generator=gen)
        H1 = hdn_amd.DLT_solve(src.to(dev), off.to(dev)).reshape(B, 9)
        H2, _ = hdn_amd.dlt_warp(src.to(dev), off.to(dev), img.to(dev))
        same = (H1 == H2.reshape(B, 9)) | (torch.isnan(H1) & torch.isnan(H2.reshape(B, 9)))
        assert bool(same.all()), scale


def test_transformer_golden_including_nudge_branch(dev):
    g = load_golden("transformer")
    y, cond = hdn_amd.transformer(T(g["img"]).to(dev), T(g["theta"]).to(dev), (20, 33))
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-4)
    assert np.abs(y.cpu().numpy() - g["y"]).max() < 1e-5
    assert cond.dim() == 0 and cond.dtype == torch.float32 and float(cond) == float(g["cond"])  # utils.py:241
    y3, cond3 = hdn_amd.transformer(T(g["img3"]).to(dev), T(g["theta3"]).to(dev), (15, 17))
    np.testing.assert_allclose(y3.cpu().numpy(), g["y3"], rtol=0, atol=1e-4)
    assert float(cond3) == float(g["cond3"])
    with pytest.raises(ValueError):
        hdn_amd.transformer(T(g["img"]).to(dev), T(g["theta"]).to(dev), (10, 10))


def test_transform_golden(dev):
    g = load_golden("transform")
    img, H = T(g["img"]).to(dev), T(g["H"]).to(dev)
    B, _, Hh, Ww = img.shape
    M, Minv = O.norm_matrices(B)
    pidx, base = O.full_patch_indices(B, Hh, Ww)
    y = hdn_amd.transform(Hh, Ww, Minv, H, M, img, pidx.to(dev), base.to(dev))
    assert y.shape == (B, 1, Hh, Ww)
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=0, atol=1e-4)
    y_fast = hdn_amd.transform(Hh, Ww, Minv, H, M, img, pidx.to(dev), base.to(dev), assume_identity_patch=True)
    assert torch.equal(y_fast.contiguous(), y.contiguous())
    # multi-channel input goes through the same kernel (NCHW in, NHWC out)
    img3 = torch.randn(2, 3, 31, 17)
    th = torch.tensor([[[1.0, 0.1, 0.0], [0.05, 0.9, 0.1], [0.02, 0.0, 1.0]]]).repeat(2, 1, 1)
    ref, _ = O.transformer(img3, th, (31, 17))
    got, _ = hdn_amd.transformer(img3.to(dev), th.to(dev), (31, 17))
    assert got.shape == (2, 31, 17, 3)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5)


def test_dlt_warp_fused_golden_and_full_batch(dev):
    g = load_golden("homo_forward")
    Hm, warped = hdn_amd.dlt_warp(T(g["h4p"]).to(dev), T(g["x"]).to(dev), T(g["org_imgs"][:, :1]).to(dev))
    np.testing.assert_allclose(Hm.cpu().numpy(), g["H_mat"], rtol=0, atol=1e-5)
    # H agrees to ~2e-6 (float64 solve here, fp32 inverse in the reference), i.e. sampling positions to ~1e-4 px;
    # these crops are white noise (|gradient| up to ~3 per px), so the fused output is held to 5e-4 here, while the
    # warp itself is held to 1e-4 on identical H in test_transform_golden / test_transformer_golden.
    d = np.abs(warped.cpu().numpy()[:1] - g["pred_I2_d"])
    assert d.max() < 5e-4 and (d > 1e-4).mean() < 1e-3
    # B=64 (BASELINE config 2): offsets N(0, 8^2) px
    r = np.random.default_rng(64)
    img = T(r.standard_normal((64, 1, 127, 127), dtype=np.float32))
    h4p = T(np.tile(np.array([0, 0, 0, 127, 127, 127, 127, 0], np.float32), (64, 1)))
    off = T((8 * r.standard_normal((64, 8))).astype(np.float32))
    Href, wref = O.dlt_warp(h4p, off, img)
    Hm, w = hdn_amd.dlt_warp(h4p.to(dev), off.to(dev), img.to(dev))
    np.testing.assert_allclose(Hm.cpu().numpy(), Href.numpy(), rtol=0, atol=1e-5)
    # the two paths hold H to ~1e-6, so sampling positions agree to ~1e-4 px; white-noise images have
    # O(1) gradients per pixel, hence the 5e-4 bound here (smooth crops are far tighter)
    assert float((w.cpu() - wref).abs().max()) < 5e-4
    # fused == two-step (same device code path for the warp)
    M, Minv = O.norm_matrices(64)
    pidx, base = O.full_patch_indices(64, 127, 127)
    two = hdn_amd.transform(127, 127, Minv, Hm, M, img.to(dev), pidx.to(dev), base.to(dev))
    assert float((two - w).abs().max()) < 1e-5
    # one channel of a [B,2,H,W] pair read in place (hdn_dlt_warp_strided_f32: no copy of org_imgs[:, :1]); odd views still work
    pair = torch.stack([img[:, 0], torch.full_like(img[:, 0], 7.0)], dim=1).to(dev)          # channel 1 must never be sampled
    Hs, ws = hdn_amd.dlt_warp(h4p.to(dev), off.to(dev), pair[:, :1])
    assert torch.equal(ws, w) and torch.equal(Hs, Hm)
    Ht, wt = hdn_amd.dlt_warp(h4p.to(dev), off.to(dev), pair.transpose(2, 3)[:, :1].transpose(2, 3))  # same view, round-tripped strides
    assert torch.equal(wt, w)
    wide = torch.zeros(64, 1, 127, 130, device=dev)
    wide[..., :127] = img.to(dev)
    assert torch.equal(hdn_amd.dlt_warp(h4p.to(dev), off.to(dev), wide[..., :127])[1], w)      # (row pitch != W: copied first)


# --------------------------------------------------------------------------- correlation heads (§8a row 11, §8f rank 1)
@pytest.mark.parametrize("tag,circular", [("ban", False), ("circ", True)])
def test_multi_ban_fused_forward_golden(dev, tag, circular):
    """MultiBAN / MultiCircBAN with the reference's weights: one correlation launch + cached template branch."""
    from test_oracle_golden import heads_fixture
    from hdn_amd import heads as HD
    sd, zfs, xfs, cls, loc = heads_fixture(tag)
    m = (HD.MultiCircBAN if circular else HD.MultiBAN)([16, 16, 16], 2, weighted=True)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    zd, xd = [z.to(dev) for z in zfs], [x.to(dev) for x in xfs]
    c, l = m(zd, xd)
    # conv / BN / 1x1 heads run on MIOpen: 1e-4 abs on O(1) outputs
    np.testing.assert_allclose(c.cpu().numpy(), cls, rtol=0, atol=1e-4)
    np.testing.assert_allclose(l.cpu().numpy(), loc, rtol=0, atol=1e-4)
    assert X.last_variant() in ("prod_29x29_5x5", "circ13")
    # second frame, same template: cached template features, same answer; new template: cache refreshed
    c2, l2 = m(zd, [x * 1.0 for x in xd])
    assert float((c2 - c).abs().max()) <= 1e-5 and float((l2 - l).abs().max()) <= 1e-5
    zd2 = [z * 0.5 for z in zd]
    c3, _ = m(zd2, xd)
    assert float((c3 - c).abs().max()) > 1e-3
    # fused schedule == the reference's per-branch schedule on the same device
    per = [getattr(m, "box" + str(i + 2))(zd[i], xd[i]) for i in range(3)]
    cw = torch.softmax(m.cls_weight, 0)
    ref_c = sum(per[i][0] * cw[i] for i in range(3))
    assert float((ref_c - c).abs().max()) < 1e-5


@pytest.mark.parametrize("tag,circular", [("ban", False), ("circ", True)])
def test_multi_ban_fused_forward_production_width(dev, tag, circular):
    """The 256-channel heads the tracker runs (prod29 / circ13 kernels, 6 problems in one launch) end to end against
    the reference's outputs; the modules are re-created from the seed (conftest.seeded_head256)."""
    from conftest import seeded_head256
    g = load_golden("heads256")
    m, zfs, xfs = seeded_head256(tag)
    psum = sum(float(v.double().sum()) for v in m.state_dict().values())
    assert abs(psum - float(g[tag + "__param_sum"])) <= 1e-6 * abs(psum), "torch's init stream drifted: regenerate heads256.npz"
    m = m.to(dev)
    c, l = m([z.to(dev) for z in zfs], [x.to(dev) for x in xfs])
    assert X.last_variant() == ("circ13" if circular else "prod_29x29_5x5")
    # 3x3 convs over 256 channels + BN + 1x1 heads on MIOpen around the HIP correlation: 2e-4 abs on O(1) outputs
    np.testing.assert_allclose(c.cpu().numpy(), g[tag + "__cls"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(l.cpu().numpy(), g[tag + "__loc"], rtol=0, atol=2e-4)


def test_multi_ban_fused_forward_config5_search_size(dev):
    """BASELINE configs[4] through the head: 37 x 37 search features (INSTANCE_SIZE = 303) -> conv_search 35 x 35 -> the
    xcorr_cfg5_kernel (6 problems, one launch) -> 31 x 31 maps, against the reference's MultiBAN output
    (tests/golden/heads256_cfg5.npz); packed (B = 1) and module-by-module forms."""
    from conftest import seeded_head256
    g = load_golden("heads256_cfg5")
    m, zfs, xfs = seeded_head256("ban_cfg5")
    psum = sum(float(v.double().sum()) for v in m.state_dict().values())
    assert abs(psum - float(g["ban__param_sum"])) <= 1e-6 * abs(psum), "torch's init stream drifted: regenerate heads256_cfg5.npz"
    m = m.to(dev)
    zd, xd = [z.to(dev) for z in zfs], [x.to(dev) for x in xfs]
    for packed in (True, False):
        object.__setattr__(m, "_hdn_no_packed_head", not packed)
        c, l = m(zd, xd)
        assert X.last_variant() == "cfg5_35x35_5x5"
        assert c.shape == (1, 2, 31, 31) and l.shape == (1, 2, 31, 31)
        np.testing.assert_allclose(c.cpu().numpy(), g["ban__cls"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(l.cpu().numpy(), g["ban__loc"], rtol=0, atol=2e-4)
    assert getattr(m, "_hdn_packed_head", None) is not None


@pytest.mark.parametrize("tag,circular", [("ban", False), ("circ", True)])
def test_packed_head_matches_module_by_module_form(dev, tag, circular):
    """The tracker's B = 1 call takes the packed path of heads.fused_forward (BatchNorm folded, the two branches of a level as one
    convolution, the 1x1 convolutions and the weighted sum as batched matrix products): same outputs as the module-by-module
    form at the production width, against the reference's golden outputs where the fixture is B = 1, and it follows its weights."""
    from conftest import seeded_head256
    from hdn_amd import heads as HD
    m, zfs, xfs = seeded_head256(tag)
    m = m.to(dev)
    z1, x1 = [z[:1].contiguous().to(dev) for z in zfs], [x[:1].contiguous().to(dev) for x in xfs]
    c, l = m(z1, x1)
    assert getattr(m, "_hdn_packed_head", None) is not None, "the packed path did not run"
    object.__setattr__(m, "_hdn_no_packed_head", True)
    c0, l0 = m(z1, x1)
    object.__setattr__(m, "_hdn_no_packed_head", False)
    assert c.shape == c0.shape and l.shape == l0.shape
    assert float((c - c0).abs().max()) <= 2e-4 and float((l - l0).abs().max()) <= 2e-4, (float((c - c0).abs().max()), float((l - l0).abs().max()))
    g = load_golden("heads256")
    if g[tag + "__cls"].shape[0] == 1:
        np.testing.assert_allclose(c.cpu().numpy(), g[tag + "__cls"], rtol=0, atol=2e-4)
        np.testing.assert_allclose(l.cpu().numpy(), g[tag + "__loc"], rtol=0, atol=2e-4)
    # in-place weight changes bump the version counters the pack is keyed on
    with torch.no_grad():
        m.loc_scale.mul_(2.0)
    c2, l2 = m(z1, x1)
    # (the batched matrix products may run split-K kernels with atomics: run to run the last bit moves)
    assert float((l2 - 2.0 * l).abs().max()) <= 1e-4 * float(l.abs().max() + 1) and float((c2 - c).abs().max()) <= 1e-5 * float(c.abs().max() + 1)
    # an unweighted 16-channel head, and a batch of two (which takes the module-by-module form)
    torch.manual_seed(9)
    u = (HD.MultiCircBAN if circular else HD.MultiBAN)([16, 16, 16], 2, weighted=False).to(dev).eval()
    for mod in u.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.7, 1.4); mod.weight.data.uniform_(0.6, 1.3); mod.bias.data.uniform_(-0.2, 0.2)
    sz, sx = (15, 15) if circular else (7, 31)
    zz, xx = [torch.randn(2, 16, sz, sz, device=dev) for _ in range(3)], [torch.randn(2, 16, sx, sx, device=dev) for _ in range(3)]
    cb, lb = u(zz, xx)                                   # B = 2: module-by-module
    ca, la = u([z[:1].contiguous() for z in zz], [x[:1].contiguous() for x in xx])   # B = 1: packed
    assert float((ca - cb[:1]).abs().max()) <= 1e-4 and float((la - lb[:1]).abs().max()) <= 1e-4


def test_fused_forward_new_template_after_old_one_is_freed(dev):
    """Two template() calls with no forward in between, the first template freed (end of one video, init of the next):
    the cache must not hand the old template's conv_kernel features to the new one, whatever ids / addresses get reused;
    nor survive a load_state_dict()."""
    import gc
    from hdn_amd import heads as HD

    def same(a, b):   # a stale cache is off by O(1); the B = 1 path's batched matrix products may differ in the last bit run to run
        return float((a - b).abs().max()) <= 1e-5 * (1.0 + float(b.abs().max()))
    torch.manual_seed(5)
    m = HD.MultiBAN([16, 16, 16], 2, weighted=True).to(dev).eval()
    gen = torch.Generator().manual_seed(6)
    mk = lambda s: [torch.randn(1, 16, s, s, generator=gen).to(dev) for _ in range(3)]
    xs = mk(31)
    z1 = mk(7)
    c1, _ = m(z1, xs)
    ids1 = [(id(z), z.data_ptr()) for z in z1]
    for trial in range(8):
        del z1
        gc.collect()
        z1 = mk(7)  # the caching allocator is free to reuse the addresses, Python the ids
        fresh = HD.MultiBAN([16, 16, 16], 2, weighted=True).to(dev).eval()
        fresh.load_state_dict(m.state_dict())
        want, _ = fresh(z1, xs)
        got, _ = m(z1, xs)
        assert same(got, want), trial
    # in-place change of the template and reloaded weights both invalidate
    z1[0].mul_(2.0)
    fresh = HD.MultiBAN([16, 16, 16], 2, weighted=True).to(dev).eval()
    fresh.load_state_dict(m.state_dict())
    assert same(m(z1, xs)[0], fresh(z1, xs)[0])
    sd = {k: (v * 1.5 if k.endswith("conv_kernel.0.weight") else v.clone()) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    fresh.load_state_dict(sd)
    assert same(m(z1, xs)[0], fresh(z1, xs)[0])
    m.train()
    with pytest.raises(RuntimeError):
        m(z1, xs)


# --------------------------------------------------------------------------- log-polar sampler (§8f rank 2)
def test_logpolar_golden(dev):
    """Crops are raw 0..255 images, so the 1e-4 bound is relative to that amplitude (2.55e-2 abs); observed <= 2e-3
    against the fixture (host-CPU table differences on white noise) and exact against the oracle on the same host."""
    g = load_golden("logpolar")
    m = hdn_amd.STN_Polar(31)
    img, polar = T(g["small_img"]).to(dev), T(g["small_polar"]).to(dev)
    for delta, ky, kg in (([0, 0], "small_y", "small_grid"), ([0, 0.3], "small_y_rot", "small_grid_rot")):
        y, grid = m(img, polar, delta)
        assert y.shape == (2, 3, 15, 15) and grid.shape == (2, 15, 15, 2)
        # the tables come from torch's CPU exp/sin/cos, whose vector paths differ by an ulp between host CPUs: the
        # fixture (made in the build container) is matched to 1e-6 here, and bit-for-bit against the oracle on THIS
        # host in test_logpolar_vs_oracle_offsets_and_border
        np.testing.assert_allclose(grid.cpu().numpy(), g[kg], rtol=0, atol=1e-6)
        np.testing.assert_allclose(y.cpu().numpy(), g[ky], rtol=0, atol=1e-4 * 255)
        assert np.abs(y.cpu().numpy() - g[ky]).max() < 1e-3
    r = golden_rng(701)
    big = (255.0 * r.random((2, 3, 255, 255))).astype(np.float32)
    y, grid = hdn_amd.STN_Polar(255)(T(big).to(dev), T(g["prod_polar"]).to(dev))
    yv = y.cpu().numpy().reshape(-1)[g["prod_idx"]]
    assert np.abs(yv - g["prod_val"]).max() < 1e-4 * 255  # white-noise 0..255 crop x ~1e-5 px table difference
    np.testing.assert_allclose(grid.cpu().numpy()[:, ::9, ::9, :], g["prod_grid"], rtol=0, atol=1e-6)
    assert abs(y.double().sum().item() - float(g["prod_sum"])) < 1e-6 * float(g["prod_sum"])
    # STN_Polar(255) on the 127-px template crop, rotation 0.2 (ModelBuilder.update_template)
    r = golden_rng(703)
    tm = (255.0 * r.random((1, 3, 127, 127))).astype(np.float32)
    y, grid = hdn_amd.STN_Polar(255)(T(tm).to(dev), torch.zeros(1, 2, device=dev), [0, 0.2])
    assert y.shape == (1, 3, 127, 127)
    assert np.abs(y.cpu().numpy().reshape(-1)[g["tmpl_idx"]] - g["tmpl_val"]).max() < 1e-4 * 255
    np.testing.assert_allclose(grid.cpu().numpy()[:, ::9, ::9, :], g["tmpl_grid"], rtol=0, atol=1e-6)
    assert abs(y.double().sum().item() - float(g["tmpl_sum"])) < 1e-6 * float(g["tmpl_sum"])


def test_logpolar_vs_oracle_offsets_and_border(dev):
    """Large offsets push samples off the crop: border padding clamps, east/south taps at the last index are masked."""
    r = np.random.default_rng(12)
    img = T((255.0 * r.random((3, 2, 63, 63))).astype(np.float32))
    polar = T(np.array([[0, 0], [40.0, -35.0], [-100.0, 90.0]], np.float32))
    for delta in ([0, 0], [0, -1.1]):
        ref, gref = O.logpolar_sample(img, polar, delta)
        y, grid = hdn_amd.STN_Polar(63)(img.to(dev), polar.to(dev), delta)
        np.testing.assert_array_equal(grid.cpu().numpy(), gref.numpy())
        assert float((y.cpu() - ref).abs().max()) < 1e-3


# --------------------------------------------------------------------------- whole head
def _seeded_net():
    torch.manual_seed(123)
    net = hdn_amd.HomoModelBuilder().eval()
    gen = torch.Generator().manual_seed(5)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.empty_like(m.running_mean).uniform_(-0.2, 0.2, generator=gen))
            m.running_var.copy_(torch.empty_like(m.running_var).uniform_(0.8, 1.2, generator=gen))
    net.fc.weight.data.mul_(0.01)
    net.fc.bias.data.copy_(torch.tensor([3.0, -2.0, 1.5, 4.0, -3.5, 2.5, 0.5, -1.0]))
    return net


def _cfg1_data(B, seed):
    r = np.random.default_rng(seed)
    datas = []
    for _ in range(B):
        tmp = O.gray_normalise(r.integers(0, 256, (127, 127, 3)))
        sea = O.gray_normalise(r.integers(0, 256, (127, 127, 3)))
        datas.append(O.merge_pair(tmp, sea))
    st = lambda k: torch.stack([torch.Tensor(d[k]).float() for d in datas])
    return {"org_imgs": st("org_imgs"), "input_tensors": st("input_tensors"), "patch_indices": st("patch_indices"), "h4p": st("four_points")}


def test_l1_score_pair_equals_two_single_launches(dev):
    r = np.random.default_rng(9)
    for n in (1, 63, 1024, 16129, 40000):
        a, b0, b1 = (T(r.standard_normal(n, dtype=np.float32)).to(dev) for _ in range(3))
        s0, s1 = G.l1_score2(a, b0, b1, 1.0 / 16129)
        assert torch.equal(s0, G.l1_score(a, b0, 1.0 / 16129)) and torch.equal(s1, G.l1_score(a, b1, 1.0 / 16129))
        ref = (a.double() - b0.double()).abs().sum().item() / 16129
        assert abs(float(s0) - ref) <= 1e-5 * max(1.0, ref)
    with pytest.raises(ValueError):
        G.l1_score2(a, b0, b1[:5], 1.0)


def test_homo_forward_golden_post_trunk(dev):
    """HomoModelBuilder.forward with the reference trunk's output injected: every HIP stage against the fixture."""
    g = load_golden("homo_forward")
    net = hdn_amd.HomoModelBuilder().eval()
    net.ShareFeature.load_state_dict(share_sd(g, "sf__"))
    net = net.to(dev)
    xfix = T(g["x"]).to(dev)
    net.fc = torch.nn.Identity()
    net.avgpool = torch.nn.Identity()

    class Inject(torch.nn.Module):
        def forward(self, feats):
            return xfix

    net.backbone = Inject()
    data = {k: T(g[k]).to(dev) for k in ("org_imgs", "input_tensors", "h4p", "patch_indices")}
    out = net(data)
    assert set(out) == {"feature_loss", "pred_I2_d", "x", "H_mat", "patch_2_res_d", "pred_I2_CnnFeature_d", "homo_neg_loss"}
    np.testing.assert_allclose(out["H_mat"].cpu().numpy(), g["H_mat"], atol=1e-5)
    np.testing.assert_allclose(out["pred_I2_d"].cpu().numpy(), g["pred_I2_d"], atol=5e-4)  # see test_dlt_warp_fused…
    np.testing.assert_allclose(out["patch_2_res_d"].cpu().numpy(), g["patch_2_res_d"], atol=1e-4)
    np.testing.assert_allclose(out["pred_I2_CnnFeature_d"].cpu().numpy(), g["pred_I2_CnnFeature_d"], atol=5e-4)
    np.testing.assert_allclose(out["feature_loss"].cpu().numpy(), g["feature_loss"], rtol=1e-3, atol=1e-8)
    assert float(out["homo_neg_loss"]) == 0.0


def test_homo_forward_negative_sample_branch_golden(dev):
    """HomoModelBuilder.forward with per-sample if_pos / if_unsup flags (homo_model_builder.py:172-205): feature_loss over the
    positive unsupervised samples only, homo_neg_loss over the negatives, against the reference's own output
    (tests/golden/homo_forward_neg.npz; trunk output injected)."""
    g, gn = load_golden("homo_forward"), load_golden("homo_forward_neg")
    net = hdn_amd.HomoModelBuilder().eval()
    net.ShareFeature.load_state_dict(share_sd(g, "sf__"))
    net = net.to(dev)
    xfix = T(gn["x"]).to(dev)
    net.fc = torch.nn.Identity()
    net.avgpool = torch.nn.Identity()

    class Inject(torch.nn.Module):
        def forward(self, feats):
            return xfix

    net.backbone = Inject()
    data = {k: T(gn[k]).to(dev) for k in ("org_imgs", "input_tensors", "h4p", "patch_indices")}
    out = net({**data, "if_pos": T(gn["if_pos"]).to(dev), "if_unsup": T(gn["if_unsup"]).to(dev)})
    np.testing.assert_allclose(out["H_mat"].cpu().numpy(), gn["H_mat"], atol=1e-5)
    np.testing.assert_allclose(out["feature_loss"].cpu().numpy(), gn["feature_loss"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(float(out["homo_neg_loss"]), float(gn["homo_neg_loss"]), rtol=1e-6)
    out = net({**data, "if_pos": torch.tensor([0.0, 1.0, 1.0], device=dev), "if_unsup": torch.ones(3, device=dev)})
    np.testing.assert_allclose(out["feature_loss"].cpu().numpy(), gn["feature_loss_b"], rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(float(out["homo_neg_loss"]), float(gn["homo_neg_loss_b"]), rtol=1e-6)


def test_track_proj_golden_tuple(dev):
    """(H_mat, similarity_norm, similarity_norm_simi) against the tuple the reference's ModelBuilder.track_proj returned
    (tests/golden/track_proj.npz; trunk output injected), in both batch orders: the scores read sample 0 only."""
    g, tp = load_golden("homo_forward"), load_golden("track_proj")
    net = hdn_amd.HomoModelBuilder().eval()
    net.ShareFeature.load_state_dict(share_sd(g, "sf__"))
    net = net.to(dev)
    net.fc = torch.nn.Identity()
    net.avgpool = torch.nn.Identity()
    data = {k: T(g[k]) for k in ("org_imgs", "input_tensors", "h4p", "patch_indices")}
    for flip, sfx in ((False, ""), (True, "_swapped")):
        xfix = (T(tp["x"]).flip(0).contiguous() if flip else T(tp["x"])).to(dev)

        class Inject(torch.nn.Module):
            def forward(self, feats):
                return xfix

        net.backbone = Inject()
        d = {k: (v.flip(0).contiguous() if flip else v).to(dev) for k, v in data.items()}
        Hm, s, ss = net.track_proj(d, None)
        np.testing.assert_allclose(Hm.cpu().numpy(), tp["H_mat" + sfx], rtol=0, atol=1e-5)
        # scores: means of |feature differences| over 127^2 pixels, warped features within 5e-4 (white-noise crops)
        assert abs(float(s) - float(tp["similarity_norm" + sfx])) <= 1e-4, (float(s), float(tp["similarity_norm" + sfx]))
        assert abs(float(ss) - float(tp["similarity_norm_simi" + sfx])) <= 1e-5


def test_track_proj_end_to_end_corner_offsets(dev):
    """Full head incl. the PyTorch-ROCm trunk vs the CPU oracle: predicted corner displacements within 1e-4."""
    net = _seeded_net()
    data = _cfg1_data(4, 77)
    sd = {k: v.clone() for k, v in net.ShareFeature.state_dict().items()}
    with torch.no_grad():
        Href, sref, ssref, aux = O.track_proj(data, sd, lambda f: net.fc(net.avgpool(net.backbone(f)).flatten(1)))
    netd = net.to(dev)
    dd = {k: v.to(dev) for k, v in data.items()}
    from hdn_amd.homo_model import homo_stages
    st = homo_stages(netd, dd)
    x = st["x"].cpu()
    err = O.corner_error(x.numpy(), aux["x"].numpy())
    assert float((x - aux["x"]).abs().max()) <= 1e-4, float((x - aux["x"]).abs().max())
    assert err.max() <= 1e-4
    Hm, s, ss = netd.track_proj(dd, None)
    assert Hm.shape == (4, 3, 3) and s.dim() == 0
    np.testing.assert_allclose(Hm.cpu().numpy(), Href.numpy(), atol=2e-5)
    assert abs(float(s) - float(sref)) <= 1e-4 and abs(float(ss) - float(ssref)) <= 1e-4
    # BN-folded channels-last trunk (§8f rank 4): same corner offsets within the north-star bound
    netd.optimize_for_inference()
    x_fast = homo_stages(netd, dd)["x"].cpu()
    assert float((x_fast - aux["x"]).abs().max()) <= 1e-4
    netd.optimize_for_inference(False)
    # cached template features give the same answer
    Hm2, s2, ss2 = netd.track_proj(dd, None, cached_patch_1=st["patch_1"].contiguous())
    assert float((Hm2 - Hm).abs().max()) < 1e-5 and abs(float(s2) - float(s)) < 1e-5


def test_config3_per_gpu_share_full_forward(dev):
    """BASELINE configs[2]: B = 512 over 8 GPUs = 64 pairs per GPU through the full HomoModelBuilder.forward (ShareFeature x2 ->
    ResNet-34 trunk -> DLT -> warp -> ShareFeature), SURVEY 8d cfg 3: parity against the CPU oracle on a 16-pair sample,
    independence of the pairs (a pair's result does not depend on its batch), and the dict contract at that batch."""
    net = _seeded_net()
    data = _cfg1_data(64, 4040)
    sd = {k: v.clone() for k, v in net.ShareFeature.state_dict().items()}
    sample = {k: v[:16] for k, v in data.items()}
    with torch.no_grad():
        ref = O.homo_forward(sample, sd, lambda f: net.fc(net.avgpool(net.backbone(f)).flatten(1)))
    netd = net.to(dev)
    dd = {k: v.to(dev) for k, v in data.items()}
    out = netd(dd)
    assert out["x"].shape == (64, 8) and out["H_mat"].shape == (64, 3, 3) and out["pred_I2_d"].shape == (1, 1, 127, 127)
    x = out["x"].cpu()
    assert float((x[:16] - ref["x"]).abs().max()) <= 1e-4          # north-star bound on the corner offsets
    assert O.corner_error(x[:16].numpy(), ref["x"].numpy()).max() <= 1e-4
    np.testing.assert_allclose(out["H_mat"][:16].cpu().numpy(), ref["H_mat"].numpy(), atol=2e-5)
    with torch.no_grad():
        net.cpu()
        fl = O.homo_forward(data, sd, lambda f: net.fc(net.avgpool(net.backbone(f)).flatten(1)))["feature_loss"].numpy()
        net.to(dev)
    np.testing.assert_allclose(out["feature_loss"].cpu().numpy(), fl, rtol=2e-3)
    # a shard of the batch gives the same offsets as the same pairs inside the full batch (what sharding over ranks relies on)
    part = netd({k: v[40:56].contiguous() for k, v in dd.items()})
    assert float((part["x"] - out["x"][40:56]).abs().max()) <= 5e-5
    assert bool(torch.isfinite(out["x"]).all()) and bool(torch.isfinite(out["H_mat"]).all())


def test_refine_warp_vs_oracle_restatement(dev):
    """hdn_refine_warp_f32 (the restated cv2.warpPerspective + H bookkeeping of hdn_tracker_proj_e2e.py:246-250) against
    the oracle's numpy restatement of the same algorithm.  PARITY UNPINNED against OpenCV itself (absent here)."""
    r = np.random.default_rng(21)
    B = 7
    img = r.standard_normal((B, 1, 127, 127)).astype(np.float32)
    src = np.tile(np.array([0, 0, 0, 127, 127, 127, 127, 0], np.float32), (B, 1))
    off = (r.standard_normal((B, 8)) * np.array([0, 1, 3, 6, 10, 16, 30])[:, None]).astype(np.float32)
    Hm = hdn_amd.DLT_solve(T(src).to(dev), T(off).to(dev)).squeeze(1)
    Hc = torch.eye(3, dtype=torch.float64, device=dev).repeat(B, 1, 1).contiguous()
    Hc[3] *= 2.0
    Hc0 = Hc.cpu().numpy().copy()
    w = hdn_amd.refine_warp(Hm, T(img).to(dev), Hc)
    w2 = hdn_amd.refine_warp(Hm, T(img).to(dev), None)
    assert torch.equal(w, w2)
    Hn = Hm.cpu().numpy()
    n_diff = 0
    for b in range(B):
        wr, Hr = O.refine_step(Hn[b], img[b, 0], Hc0[b])
        np.testing.assert_allclose(Hc[b].cpu().numpy(), Hr, rtol=1e-6, atol=1e-7)
        d = np.abs(w[b, 0].cpu().numpy() - wr)
        # identical 1/32-px coordinates => identical taps and weights; a float32 inverse that differs in its last bit can
        # move a coordinate across a rounding tie for a few pixels, each by 1/32 px of a white-noise image
        n_diff += int((d > 1e-5).sum())
        assert d.max() < 0.25, (b, d.max())
    assert n_diff <= 16, n_diff
    # identity H leaves the crop untouched; out-of-range samples replicate the border
    eye = torch.eye(3, device=dev).repeat(B, 1, 1)
    assert torch.equal(hdn_amd.refine_warp(eye, T(img).to(dev)), T(img).to(dev))
    sh = eye.clone(); sh[:, 0, 2] = 40.0
    ws = hdn_amd.refine_warp(sh, T(img).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(ws[:, 0, :, :40], np.repeat(img[:, 0, :, :1], 40, axis=2))
    with pytest.raises(ValueError):
        hdn_amd.refine_warp(eye[:2], T(img).to(dev))


def test_homo_refine_two_iterations_config5(dev):
    """BASELINE configs[4]'s "2-scale coarse-to-fine": the refinement loop with trip count 2, device-resident, against the
    oracle's loop (same seeded head; the trunk runs on MIOpen vs PyTorch-CPU), then at the config's batch of 256."""
    net = _seeded_net()
    B = 3
    d = _cfg1_data(B, 909)
    tmpl, srch = d["org_imgs"][:, :1].contiguous(), d["org_imgs"][:, 1:].contiguous()
    sd = {k: v.clone() for k, v in net.ShareFeature.state_dict().items()}
    with torch.no_grad():
        Hc_ref, s_ref, ss_ref, Hs = O.homo_refine(tmpl, srch, sd, lambda f: net.fc(net.avgpool(net.backbone(f)).flatten(1)), 2)
    netd = net.to(dev)
    Hc, s, ss = hdn_amd.homo_refine(netd, tmpl.to(dev), srch.to(dev), iterations=2)
    assert Hc.dtype == torch.float64 and Hc.shape == (B, 3, 3)
    # H entries: offsets within 1e-4 px (north-star bound) => H within ~1e-5; second iteration sees a crop resampled at
    # 1/32-px coordinates that may differ at a few ties
    np.testing.assert_allclose(Hc.cpu().numpy(), Hc_ref, rtol=0, atol=2e-4)
    assert abs(float(s) - float(s_ref)) <= 2e-4 and abs(float(ss) - float(ss_ref)) <= 1e-5
    # one iteration == the shipped tracker's loop: H_comp = inv(H)/inv(H)[2,2]
    Hc1, _, _ = hdn_amd.homo_refine(netd, tmpl.to(dev), srch.to(dev), iterations=1)
    Hi = np.linalg.inv(Hs[0].numpy().astype(np.float64))
    np.testing.assert_allclose(Hc1.cpu().numpy(), Hi / Hi[:, 2:3, 2:3], rtol=0, atol=1e-4)
    # the config's batch: 256 pairs in one call, pairs independent (row b of a batched call == that pair alone)
    big = _cfg1_data(8, 910)
    t8, s8 = big["org_imgs"][:, :1].to(dev), big["org_imgs"][:, 1:].to(dev)
    tt, st = t8.repeat(32, 1, 1, 1), s8.repeat(32, 1, 1, 1)
    Hb, _, _ = hdn_amd.homo_refine(netd, tt, st, iterations=2)
    assert Hb.shape == (256, 3, 3) and bool(torch.isfinite(Hb).all())
    assert float((Hb[:8] - Hb[248:]).abs().max()) <= 1e-4
    H8, _, _ = hdn_amd.homo_refine(netd, t8, s8, iterations=2)
    assert float((Hb[:8] - H8).abs().max()) <= 2e-4
    # the two forms of the loop body: separate crops + cached ShareFeature(template) (track_proj_pair, the tracker's path) and the
    # reference-shaped data dictionary without the cache give the same H and scores
    Hp, sp, ssp = hdn_amd.homo_refine(netd, t8, s8, iterations=2, cache_template=True)
    Hd, sd_, ssd = hdn_amd.homo_refine(netd, t8, s8, iterations=2, cache_template=False)
    assert float((Hp - Hd).abs().max()) <= 2e-4 and abs(float(sp) - float(sd_)) <= 1e-5 and abs(float(ssp) - float(ssd)) <= 1e-6
    # ... and with ShareFeature(template) handed in (what the tracker keeps for the whole sequence): the same launches minus one
    with torch.no_grad():
        p1 = netd.ShareFeature(t8)
    Hq, sq, ssq = hdn_amd.homo_refine(netd, t8, s8, iterations=2, patch_1=p1)
    # (this network's trunk runs on MIOpen, whose convolutions are not bit-reproducible run to run: the bounds of the comparison above)
    assert float((Hq - Hp).abs().max()) <= 2e-4 and abs(float(sq) - float(sp)) <= 1e-5 and abs(float(ssq) - float(ssp)) <= 1e-6
    with pytest.raises(ValueError):
        hdn_amd.homo_refine(netd, t8, s8, iterations=1, patch_1=p1[:4])
    from hdn_amd.homo_model import track_proj_pair
    with pytest.raises(ValueError):
        track_proj_pair(netd, t8, s8[:, :, :100], None, None)


def test_graphed_track_proj_matches_eager(dev):
    """hipGraph replay of the per-frame head (B=1) gives the eager result on new inputs (static buffers refreshed)."""
    from hdn_amd.graph import GraphedTrackProj
    net = _seeded_net().to(dev)
    frames = [{k: v.to(dev) for k, v in _cfg1_data(1, 500 + t).items()} for t in range(3)]
    # same template in every frame (as in a tracked sequence)
    for f in frames[1:]:
        f["org_imgs"][:, :1] = frames[0]["org_imgs"][:, :1]
        f["input_tensors"][:, :1] = frames[0]["input_tensors"][:, :1]
    gr = GraphedTrackProj(net, frames[0], template_constant=True)
    for f in frames:
        He, se, sse = net.track_proj(f, None)
        Hg, sg, ssg = gr(f)
        assert float((He - Hg).abs().max()) < 1e-5
        assert abs(float(se) - float(sg)) < 1e-5 and abs(float(sse) - float(ssg)) < 1e-5


# --------------------------------------------------------------------------- trunk, first stage
# (batches below ~1,000 waves take the 1-row-strip kernel, the two large ones the 4-row-strip kernel)
@pytest.mark.parametrize("shape", [(2, 127, 127), (1, 9, 13), (3, 64, 128), (2, 5, 7), (1, 126, 125), (1, 2, 2), (1, 4, 1), (1, 3, 130), (33, 127, 127),
                                   (130, 31, 29)])
def test_trunk_fused_stem_vs_torch(dev, shape):
    """hdn_trunk_stem_f32 (conv 7x7/s2 + folded BN + ReLU + maxpool 3x3/s2 in one kernel) against the same stage of the folded trunk
    in PyTorch on the CPU; both memory layouts; widths outside the kernel's range take the library path inside FusedStem."""
    from hdn_amd.trunk import FusedStem, fold_for_inference, resnet34_homo
    torch.manual_seed(5)
    base = resnet34_homo().eval()
    for m in base.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
    cpu = fold_for_inference(base, channels_last=False)
    B, H, W = shape
    x = torch.randn(B, 2, H, W)
    with torch.no_grad():
        ref = cpu.maxpool(cpu.relu(cpu.conv1(x)))
    for nhwc in (False, True):
        st = FusedStem(cpu.conv1, nhwc).to(dev)
        y = st(x.to(dev))
        assert y.shape == ref.shape
        if 2 <= W <= 128:
            assert y.is_contiguous(memory_format=torch.channels_last if nhwc else torch.contiguous_format)
        assert float((y.cpu() - ref).abs().max()) <= 2e-5 + 2e-6 * float(ref.abs().max())
    with pytest.raises(ValueError):
        FusedStem(torch.nn.Conv2d(2, 64, 7, 2, 3, bias=False), False)


@pytest.mark.parametrize("B", [1, 5, 8, 13, 37, 48, 64])
def test_trunk_stem_matrix_core_form_vs_float64(dev, B):
    """hdn_trunk_stem_mfma_f32 (the first stage as an implicit GEMM on the matrix cores, fp32 as two fp16 pieces; every batch size at
    127 px, channels-last; 4 conv rows per workgroup below 8 images, 8 below 48, 16 from there on) against the same stage in float64 on the CPU, and against the vector-pipe kernel it stands in for: the error of
    an fp32 convolution.  Inputs at the tracker's scale (crops are 0..255 minus a mean) as well as unit normal."""
    from hdn_amd import trunk
    from hdn_amd.trunk import FusedStem
    g = torch.Generator().manual_seed(900 + B)
    conv = torch.nn.Conv2d(2, 64, 7, 2, 3)
    conv.weight.data = torch.randn(64, 2, 7, 7, generator=g) * 0.1
    conv.bias.data = torch.randn(64, generator=g) * 0.5
    st = FusedStem(conv, True).to(dev)
    assert B >= trunk.STEM_MFMA_MIN_BATCH
    for scale in (1.0, 120.0):
        x = torch.randn(B, 2, 127, 127, generator=g) * scale
        x[0, :, :, :5] = scale; x[B - 1, :, -4:, :] = -scale            # (edges: the zero padding must not leak)
        with torch.no_grad():
            ref = F.max_pool2d(F.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), stride=2, padding=3)), 3, 2, 1)
        y = st(x.to(dev))
        assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
        st.mfma_disabled = True
        y_valu = st(x.to(dev))
        st.mfma_disabled = False
        mag = float(ref.abs().max())
        e, e_valu = float((y.cpu().double() - ref).abs().max()), float((y_valu.cpu().double() - ref).abs().max())
        assert e <= 3e-6 * mag, (e, mag)
        assert e <= 4 * e_valu + 1e-7 * mag, (e, e_valu)                 # no worse than the fp32 kernel's own rounding, to a small factor


def test_folded_trunk_with_and_without_fused_stem(dev):
    from hdn_amd.trunk import fold_for_inference, resnet34_homo
    torch.manual_seed(6)
    base = resnet34_homo().eval().to(dev)
    x = torch.randn(4, 2, 127, 127, device=dev)
    with torch.no_grad():
        ref = base(x)
        for cl in (False, True):
            a = fold_for_inference(base, channels_last=cl, fused_stem=False)(x.contiguous(memory_format=torch.channels_last) if cl else x)
            b = fold_for_inference(base, channels_last=cl, fused_stem=True)(x)
            scale = float(ref.abs().max())
            assert float((a - ref).abs().max()) <= 2e-4 * scale and float((b - ref).abs().max()) <= 2e-4 * scale
            assert float((a - b).abs().max()) <= 1e-4 * scale


# --------------------------------------------------------------------------- trunk epilogues + the benchmarked configuration
@pytest.mark.parametrize("shape,nhwc", [((3, 64, 8, 8), True), ((2, 12, 5, 7), True), ((2, 6, 5, 7), True), ((3, 10, 4, 6), False),
                                         ((2, 5, 3, 3), False), ((64, 64, 32, 32), True), ((64, 512, 4, 4), True), ((2, 128, 16, 16), False)])
@pytest.mark.parametrize("with_res", [False, True])
def test_bias_relu_epilogue_bitexact(dev, shape, nhwc, with_res):
    """hdn_bias_relu_f32 (every kernel instantiation: 16-byte NHWC with power-of-two and other channel counts, 16-byte NCHW,
    the one-float-per-lane form for layouts that do not vectorise; with and without residual) against PyTorch's own
    relu((y + b) + r): same operations in the same order, so bit-exact."""
    from hdn_amd.trunk import bias_relu_
    g = torch.Generator().manual_seed(sum(shape) + 7 * nhwc + with_res)
    y = torch.randn(shape, generator=g)
    b = torch.randn(shape[1], generator=g)
    r = torch.randn(shape, generator=g) if with_res else None
    ref = y + b.view(1, -1, 1, 1)
    if with_res:
        ref = ref + r
    ref = torch.relu(ref)
    fmt = torch.channels_last if nhwc else torch.contiguous_format
    yd = y.to(dev).contiguous(memory_format=fmt)
    rd = r.to(dev).contiguous(memory_format=fmt) if with_res else None
    out = bias_relu_(yd, b.to(dev), rd)
    assert out.data_ptr() == yd.data_ptr()
    assert torch.equal(out.cpu(), ref)
    with pytest.raises(ValueError):
        bias_relu_(yd, b.to(dev)[:-1])
    if with_res:
        with pytest.raises(ValueError):     # HDN_E_ALIAS
            bias_relu_(yd, b.to(dev), yd)


@pytest.mark.parametrize("C,S", [(64, 32), (128, 16), (256, 8), (512, 4)])
@pytest.mark.parametrize("B", [1, 3, 64])
def test_conv3x3_matrix_core_vs_float64(dev, C, S, B):
    """hdn_conv3x3_bias_relu_f32 (split-bf16 implicit GEMM on the matrix cores, fused bias / residual / ReLU) against a float64
    convolution: its error must stay within 4x the error of PyTorch's own fp32 convolution + 1e-5 of the output scale (observed
    3-8e-6 on outputs of O(5); fp32: 1e-6), odd and single batch sizes included (ragged last tiles)."""
    import torch.nn.functional as F
    from hdn_amd.trunk import pack_conv3x3, conv3x3_bias_relu
    g = torch.Generator().manual_seed(C + B)
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    nb = min(B, 5)                                      # the float64 truth on the first and last images only
    x = torch.randn(B, C, S, S, generator=g).clamp_min_(0)
    r = torch.randn(B, C, S, S, generator=g)
    cl = torch.channels_last
    wp, bd = pack_conv3x3(w).to(dev), b.to(dev)
    xd, rd = x.to(dev).contiguous(memory_format=cl), r.to(dev).contiguous(memory_format=cl)
    y = conv3x3_bias_relu(xd, wp, bd, rd)
    y0 = conv3x3_bias_relu(xd, wp, bd)
    assert torch.equal(y, conv3x3_bias_relu(xd, wp, bd, rd))          # deterministic
    assert y.is_contiguous(memory_format=cl) and y.shape == x.shape
    for sl in (slice(0, nb), slice(B - nb, B)):
        conv = F.conv2d(x[sl].double(), w.double(), b.double(), padding=1)
        t, t0 = torch.relu(conv + r[sl].double()), torch.relu(conv)
        ref = torch.relu(F.conv2d(x[sl], w, b, padding=1) + r[sl])
        e_ref = float((ref.double() - t).abs().max())
        scale = float(t.abs().max())
        for got, truth in ((y, t), (y0, t0)):
            e = float((got[sl].cpu().double() - truth).abs().max())
            assert e <= 4 * e_ref + 1e-5 * scale, (e, e_ref, scale)
    with pytest.raises(ValueError):
        conv3x3_bias_relu(xd.contiguous(), wp, bd)                    # NCHW input
    with pytest.raises(ValueError):
        conv3x3_bias_relu(xd, wp[:-1], bd)


@pytest.mark.parametrize("C,S", [(64, 32), (128, 16), (256, 8), (512, 4)])
@pytest.mark.parametrize("B", [24, 33, 64])
def test_conv3x3_v2_large_batch_form_vs_float64(dev, C, S, B):
    """hdn_conv3x3_v2_f32 (round 5: 64 x 64 tiles per consumer wave, K split over the consumers, weights streamed L2 -> registers in
    fragment order) against a float64 convolution — same bound as hdn_conv3x3_bias_relu_f32 (4x the error of PyTorch's own fp32
    convolution + 1e-5 of the output scale) — and against that kernel (same arithmetic, different order of the fp32 partial sums);
    ragged batches (33: partial last tiles; 24 / 33: K split over workgroups + reduction launch at several shapes), determinism."""
    import torch.nn.functional as F
    from hdn_amd.trunk import pack_conv3x3, pack_conv3x3_v2, conv3x3_bias_relu, V2_MIN_BATCH
    assert B >= V2_MIN_BATCH
    g = torch.Generator().manual_seed(7 * C + B)
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    nb = 3
    x = torch.randn(B, C, S, S, generator=g).clamp_min_(0)
    r = torch.randn(B, C, S, S, generator=g)
    cl = torch.channels_last
    wp, wp2, bd = pack_conv3x3(w).to(dev), pack_conv3x3_v2(w).to(dev), b.to(dev)
    xd, rd = x.to(dev).contiguous(memory_format=cl), r.to(dev).contiguous(memory_format=cl)
    y = conv3x3_bias_relu(xd, wp, bd, rd, wpacked_v2=wp2)
    y0 = conv3x3_bias_relu(xd, wp, bd, wpacked_v2=wp2)
    assert torch.equal(y, conv3x3_bias_relu(xd, wp, bd, rd, wpacked_v2=wp2))          # deterministic
    assert y.is_contiguous(memory_format=cl) and y.shape == x.shape
    y1 = conv3x3_bias_relu(xd, wp, bd, rd)                                            # the round-4 kernel
    scale_all = float(y1.abs().max())
    assert float((y - y1).abs().max()) <= 2e-6 * scale_all + 1e-6, float((y - y1).abs().max())
    for sl in (slice(0, nb), slice(B - nb, B)):
        conv = F.conv2d(x[sl].double(), w.double(), b.double(), padding=1)
        t, t0 = torch.relu(conv + r[sl].double()), torch.relu(conv)
        ref = torch.relu(F.conv2d(x[sl], w, b, padding=1) + r[sl])
        e_ref = float((ref.double() - t).abs().max())
        scale = float(t.abs().max())
        for got, truth in ((y, t), (y0, t0)):
            e = float((got[sl].cpu().double() - truth).abs().max())
            assert e <= 4 * e_ref + 1e-5 * scale, (e, e_ref, scale)
    with pytest.raises(ValueError):
        conv3x3_bias_relu(xd, wp, bd, wpacked_v2=wp2[:-1])


@pytest.mark.parametrize("CI,S", [(64, 16), (128, 8), (256, 4)])
@pytest.mark.parametrize("B", [1, 3, 64])
def test_conv3x3s2_ds_matrix_core_vs_float64(dev, CI, S, B):
    """hdn_conv3x3s2_ds_f32: the stride-2 convolution (+ bias, ReLU) and the block's 1x1 / stride-2 downsample branch from one
    staged input, against float64 convolutions; same bound as the stride-1 kernel."""
    import torch.nn.functional as F
    from hdn_amd.trunk import pack_conv3x3s2_ds, conv3x3s2_ds
    g = torch.Generator().manual_seed(CI + B)
    CO = 2 * CI
    w = torch.randn(CO, CI, 3, 3, generator=g) * (2.0 / (9 * CI)) ** 0.5
    wd = torch.randn(CO, CI, 1, 1, generator=g) * (1.0 / CI) ** 0.5
    b = torch.randn(CO, generator=g) * 0.1
    x = torch.randn(B, CI, 2 * S, 2 * S, generator=g).clamp_min_(0)
    cl = torch.channels_last
    wp = pack_conv3x3s2_ds(w, wd).to(dev)
    xd = x.to(dev).contiguous(memory_format=cl)
    y, yd = conv3x3s2_ds(xd, wp, b.to(dev))
    y2, yd2 = conv3x3s2_ds(xd, wp, b.to(dev))
    assert torch.equal(y, y2) and torch.equal(yd, yd2) and y.shape == (B, CO, S, S) and yd.is_contiguous(memory_format=cl)
    nb = min(B, 5)
    for sl in (slice(0, nb), slice(B - nb, B)):
        t = torch.relu(F.conv2d(x[sl].double(), w.double(), b.double(), stride=2, padding=1))
        td = F.conv2d(x[sl].double(), wd.double(), None, stride=2)
        ref = torch.relu(F.conv2d(x[sl], w, b, stride=2, padding=1))
        refd = F.conv2d(x[sl], wd, None, stride=2)
        for got, truth, r32 in ((y, t, ref), (yd, td, refd)):
            e, e_ref, scale = float((got[sl].cpu().double() - truth).abs().max()), float((r32.double() - truth).abs().max()), float(truth.abs().max())
            assert e <= 4 * e_ref + 1e-5 * scale, (e, e_ref, scale)


@pytest.mark.parametrize("CI,S", [(64, 16), (128, 8), (256, 4)])
@pytest.mark.parametrize("B", [24, 33, 64])
def test_conv3x3s2_v2_large_batch_form_vs_float64(dev, CI, S, B):
    """hdn_conv3x3s2_v2_f32 (round 5: the stride-2 stages in the producer / consumer form, even / odd column images, the downsample branch as a
    tenth tap step) against float64 convolutions and against hdn_conv3x3s2_ds_f32: the same bound as every other convolution kernel here; partial
    tiles (B = 33: the 4 x 4 stage's tiles hold four images), borders, both outputs; bit-reproducible."""
    from hdn_amd.trunk import pack_conv3x3s2_ds, pack_conv3x3s2_ds_v2, conv3x3s2_ds, V2_MIN_BATCH
    assert B >= V2_MIN_BATCH
    g = torch.Generator().manual_seed(3 * CI + B)
    CO = 2 * CI
    w = torch.randn(CO, CI, 3, 3, generator=g) * (2.0 / (9 * CI)) ** 0.5
    wd = torch.randn(CO, CI, 1, 1, generator=g) * (1.0 / CI) ** 0.5
    b = torch.randn(CO, generator=g) * 0.1
    x = torch.randn(B, CI, 2 * S, 2 * S, generator=g).clamp_min_(0)
    x[0, :, 0, :] = 3.0; x[B - 1, :, :, 2 * S - 1] = 2.0                 # (borders: the padding row / column must stay zero)
    cl = torch.channels_last
    wp, wp2 = pack_conv3x3s2_ds(w, wd).to(dev), pack_conv3x3s2_ds_v2(w, wd).to(dev)
    xd = x.to(dev).contiguous(memory_format=cl)
    y, yd = conv3x3s2_ds(xd, wp, b.to(dev), wpacked_v2=wp2)
    y2, yd2 = conv3x3s2_ds(xd, wp, b.to(dev), wpacked_v2=wp2)
    y1, yd1 = conv3x3s2_ds(xd, wp, b.to(dev))
    assert torch.equal(y, y2) and torch.equal(yd, yd2) and y.shape == (B, CO, S, S) and yd.is_contiguous(memory_format=cl)
    nb = min(B, 5)
    for sl in (slice(0, nb), slice(B - nb, B)):
        t = torch.relu(F.conv2d(x[sl].double(), w.double(), b.double(), stride=2, padding=1))
        td = F.conv2d(x[sl].double(), wd.double(), None, stride=2)
        ref = torch.relu(F.conv2d(x[sl], w, b, stride=2, padding=1))
        refd = F.conv2d(x[sl], wd, None, stride=2)
        for got, got1, truth, r32 in ((y, y1, t, ref), (yd, yd1, td, refd)):
            e, e_ref, scale = float((got[sl].cpu().double() - truth).abs().max()), float((r32.double() - truth).abs().max()), float(truth.abs().max())
            assert e <= 4 * e_ref + 1e-5 * scale, (e, e_ref, scale)
            assert float((got[sl] - got1[sl]).abs().max()) <= 2e-6 * scale        # the round-4 kernel: another summation order, same pieces
    with pytest.raises(ValueError):
        conv3x3s2_ds(xd, wp, b.to(dev), wpacked_v2=wp2[:-8])


@pytest.mark.parametrize("C,S", [(64, 32), (128, 16), (256, 8), (512, 4)])
@pytest.mark.parametrize("B", [1, 2, 3, 9, 16])
def test_conv3x3_chain_vs_unchained_and_float64(dev, C, S, B):
    """hdn_conv3x3_chain_f32 / hdn_conv3x3_finish_f32: two BasicBlocks as four chained launches (each convolution finishes the
    previous one's raw K slices while it stages them; the second block's residual is the activation the first launch of that block
    wrote out on the way) against the same blocks through hdn_conv3x3_bias_relu_f32 — bit-identical whenever both forms split K the same
    way (they do at small B) — and against float64."""
    import ctypes
    import torch.nn.functional as F
    from hdn_amd import _lib
    from hdn_amd.trunk import pack_conv3x3, conv3x3_bias_relu, chain_conv, LazyAct
    g = torch.Generator().manual_seed(7 * C + B)
    cl = torch.channels_last
    ws = [torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5 for _ in range(4)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(4)]
    x = torch.randn(B, C, S, S, generator=g).clamp_min_(0)
    wp, bd = [pack_conv3x3(w).to(dev) for w in ws], [b.to(dev) for b in bs]
    xd = x.to(dev).contiguous(memory_format=cl)
    # unchained
    y1 = conv3x3_bias_relu(conv3x3_bias_relu(xd, wp[0], bd[0]), wp[1], bd[1], xd)
    y2 = conv3x3_bias_relu(conv3x3_bias_relu(y1, wp[2], bd[2]), wp[3], bd[3], y1)
    # chained
    s1, none, xo = chain_conv(xd, wp[0], 1, want_x=True)
    assert none is None and xo is None
    s2, _, _ = chain_conv(LazyAct(s1, bd[0]), wp[1])
    l1 = LazyAct(s2, bd[1], xd)
    s3, _, y1c = chain_conv(l1, wp[2], 1, want_x=True)
    s4, _, _ = chain_conv(LazyAct(s3, bd[2]), wp[3])
    y2c = LazyAct(s4, bd[3], y1c).finish()
    lib = _lib.load()
    same_split = lib.hdn_conv3x3_chain_slices(B, S, C, 1) * B * S * S * C * 4 == max(lib.hdn_conv3x3_workspace_bytes(B, S, C, 1), B * S * S * C * 4)
    assert y1c.shape == y1.shape and y1c.is_contiguous(memory_format=cl) and y2c.is_contiguous(memory_format=cl)
    assert torch.equal(y1c, l1.finish())                   # the activation written on the way == the finishing launch
    if same_split:
        assert torch.equal(y1c, y1) and torch.equal(y2c, y2)
    assert torch.equal(y2c, LazyAct(chain_conv(LazyAct(s3, bd[2]), wp[3])[0], bd[3], y1c).finish())   # deterministic
    xt = x.double()
    t1 = torch.relu(F.conv2d(torch.relu(F.conv2d(xt, ws[0].double(), bs[0].double(), padding=1)), ws[1].double(), bs[1].double(), padding=1) + xt)
    t2 = torch.relu(F.conv2d(torch.relu(F.conv2d(t1, ws[2].double(), bs[2].double(), padding=1)), ws[3].double(), bs[3].double(), padding=1) + t1)
    for got, truth in ((y1c, t1), (y2c, t2), (y2, t2)):
        assert float((got.cpu().double() - truth).abs().max()) <= 2e-5 * float(truth.abs().max())
    one = ctypes.c_void_p(64)
    assert lib.hdn_conv3x3_chain_f32(one, 0, None, None, 0, None, one, None, None, 1, S, C, 1, 0, None) == -1          # no output
    assert lib.hdn_conv3x3_chain_f32(one, 2, None, None, 0, None, one, ctypes.c_void_p(128), None, 1, S, C, 1, 0, None) == -1   # slices without their bias
    assert lib.hdn_conv3x3_chain_f32(one, 0, None, one, 1, None, one, ctypes.c_void_p(128), None, 1, S, C, 1, 0, None) == -2    # an activation with a residual
    assert lib.hdn_conv3x3_chain_f32(one, 0, None, None, 0, None, one, one, None, 1, S, C, 1, 0, None) == -4                   # in place
    assert lib.hdn_conv3x3_finish_f32(one, 0, one, None, 0, ctypes.c_void_p(128), 1, S, C, None) == -2


@pytest.mark.parametrize("CI,S", [(64, 16), (128, 8), (256, 4)])
@pytest.mark.parametrize("B", [1, 3])
def test_conv3x3_chain_downsample_block(dev, CI, S, B):
    """The first block of layer2..4 chained: stride-2 convolution + downsample branch from a LazyAct input, the second convolution
    from the first one's slices, the block's output with the downsample SLICES as its residual — against the unchained launches
    (bit-identical at these batch sizes) and float64."""
    import torch.nn.functional as F
    from hdn_amd.trunk import pack_conv3x3, pack_conv3x3s2_ds, conv3x3s2_ds, conv3x3_bias_relu, chain_conv, LazyAct
    g = torch.Generator().manual_seed(11 * CI + B)
    cl = torch.channels_last
    CO = 2 * CI
    w0 = torch.randn(CI, CI, 3, 3, generator=g) * (2.0 / (9 * CI)) ** 0.5
    w1 = torch.randn(CO, CI, 3, 3, generator=g) * (2.0 / (9 * CI)) ** 0.5
    wd = torch.randn(CO, CI, 1, 1, generator=g) * (1.0 / CI) ** 0.5
    w2 = torch.randn(CO, CO, 3, 3, generator=g) * (2.0 / (9 * CO)) ** 0.5
    b0, b1, b2 = (torch.randn(c, generator=g) * 0.1 for c in (CI, CO, CO))
    x = torch.randn(B, CI, 2 * S, 2 * S, generator=g).clamp_min_(0)
    r = torch.randn(B, CI, 2 * S, 2 * S, generator=g)
    xd, rd = x.to(dev).contiguous(memory_format=cl), r.to(dev).contiguous(memory_format=cl)
    p0, p1, p2 = pack_conv3x3(w0).to(dev), pack_conv3x3s2_ds(w1, wd).to(dev), pack_conv3x3(w2).to(dev)
    b0d, b1d, b2d = b0.to(dev), b1.to(dev), b2.to(dev)
    # unchained: a = relu(conv(x, w0) + b0 + r); (y, idt) = s2 block; out = relu(conv(y, w2) + b2 + idt)
    a = conv3x3_bias_relu(xd, p0, b0d, rd)
    y, idt = conv3x3s2_ds(a, p1, b1d)
    out = conv3x3_bias_relu(y, p2, b2d, idt)
    # chained
    s0, _, _ = chain_conv(xd, p0)
    s1, sd, _ = chain_conv(LazyAct(s0, b0d, rd), p1, 2)
    s2, _, _ = chain_conv(LazyAct(s1, b1d), p2)
    outc = LazyAct(s2, b2d, sd).finish()
    assert sd.dim() == 5 and sd.shape == s1.shape and outc.shape == out.shape
    assert torch.equal(outc, out)
    # ... and into the next block: the downsample slices as the residual of a LAZY input
    w3 = torch.randn(CO, CO, 3, 3, generator=g) * (2.0 / (9 * CO)) ** 0.5
    p3 = pack_conv3x3(w3).to(dev)
    s3, _, xo = chain_conv(LazyAct(s2, b2d, sd), p3, 1, want_x=True)
    assert torch.equal(xo, out)
    assert torch.equal(LazyAct(s3, b1d).finish(), conv3x3_bias_relu(out, p3, b1d))
    at = torch.relu(F.conv2d(x.double(), w0.double(), b0.double(), padding=1) + r.double())
    yt = torch.relu(F.conv2d(at, w1.double(), b1.double(), stride=2, padding=1))
    tt = torch.relu(F.conv2d(yt, w2.double(), b2.double(), padding=1) + F.conv2d(at, wd.double(), None, stride=2))
    assert float((outc.cpu().double() - tt).abs().max()) <= 2e-5 * float(tt.abs().max())


def test_trunk_scaled_activation_domain_is_exact(dev, monkeypatch):
    """A fully fused trunk runs its interior in the scaled activation domain (hdn_amd.trunk.ACT_SCALE_LOG2: the first stage writes relu(conv) * 2^-8,
    every block takes and gives x * 2^-8 with biases * 2^-8, the exit multiplies by 2^8), so that the 2^-8 multiply of the fp16 split is paid once per
    trunk instead of once per layer.  All of those scalings are powers of two, so the result must be BIT-IDENTICAL to the same trunk with every kernel in
    real units (HDN_TRUNK_SCALED_DOMAIN=0: act_domain = 0 everywhere) — at the chained batch sizes, the single-launch ones and the large-batch form, on
    ordinary inputs and on inputs of 7e4 (beyond fp16, inside the format's 1.67e7), through forward() and through the fused avgpool + fc tail."""
    import hdn_amd
    from hdn_amd import trunk as T
    from hdn_amd.homo_model import _regress
    torch.manual_seed(9)
    net = hdn_amd.HomoModelBuilder().to(dev).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.8, 1.2)
    folds = {}
    for dom in ("1", "0"):
        monkeypatch.setenv("HDN_TRUNK_SCALED_DOMAIN", dom)
        folds[dom] = T.fold_for_inference(net.backbone, channels_last=True, fused_stem=True, fused_epilogue=True)
    a, b = folds["1"], folds["0"]
    assert a.act_domain == 1 and b.act_domain == 0 and a.conv1.out_domain == 1 and all(m.act_domain == 1 for m in a.modules() if isinstance(m, T.FusedBasicBlock))
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for B, big in ((1, 0.0), (3, 0.0), (T.CHAIN_MAX_BATCH + 2, 0.0), (T.V2_MIN_BATCH + 1, 0.0), (2, 7.0e4), (T.V2_MIN_BATCH, -7.0e4)):
            x = torch.randn(B, 2, 127, 127, generator=g)
            if big:
                x[B - 1, 1, 64, 70] = big
            x = x.to(dev)
            ya, yb = a(x), b(x)
            assert torch.isfinite(ya).all() and torch.equal(ya, yb), (B, big, float((ya - yb).abs().max()))
            sa = a.forward_scaled(x)
            assert torch.equal(sa * 256.0, yb)                                   # the interior really is x 2^-8
            # the first stage alone, and the regressor through the fused tail (hdn_avgpool_fc_f32 with in_domain = 1)
            assert torch.equal(a.conv1(x) * 256.0, b.conv1(x))
            outs = []
            for f in (a, b):
                object.__setattr__(net, "_hdn_fast_trunk", f)
                object.__setattr__(net, "_hdn_fast_nhwc", False)
                outs.append(_regress(net, x))
            assert outs[0].shape == (B, 8) and torch.equal(outs[0], outs[1])
            if not big:
                ref = net.fc(net.avgpool(net.backbone(x)).flatten(1))
                assert float((outs[0] - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    object.__setattr__(net, "_hdn_fast_trunk", None)


def test_chained_trunk_equals_unchained(dev):
    """The folded trunk at B = 1 and 2 in its chained form (37 launches instead of 70) against the same trunk with the switch off:
    bit-identical (same K split, same order of additions), eagerly and as a captured hipGraph."""
    from hdn_amd.trunk import fold_for_inference, resnet34_homo, FusedBasicBlock
    torch.manual_seed(5)
    net = resnet34_homo().to(dev).eval()
    fast = fold_for_inference(net, channels_last=True, fused_stem=True, fused_epilogue=True)
    for B in (1, 2):
        x = torch.randn(B, 2, 127, 127, device=dev).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y = fast(x)
            FusedBasicBlock.chain_disabled = True
            try:
                y0 = fast(x)
            finally:
                FusedBasicBlock.chain_disabled = False
            assert y.shape == (B, 512, 4, 4) and torch.equal(y, y0)
            ref = net(x)
            assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fast(x)
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                yg = fast(x)
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(yg, y)
    # larger chained batches (the two forms may split K differently there: a tolerance), and the first unchained one
    from hdn_amd import trunk as T
    for B in (5, T.CHAIN_MAX_BATCH, T.CHAIN_MAX_BATCH + 1):
        x = torch.randn(B, 2, 127, 127, device=dev).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y = fast(x)
            FusedBasicBlock.chain_disabled = True
            try:
                y0 = fast(x)
            finally:
                FusedBasicBlock.chain_disabled = False
            assert float((y - y0).abs().max()) <= 2e-5 * float(y0.abs().max())
            assert float((y - net(x)).abs().max()) <= 1e-4 * float(y0.abs().max())


def test_fused_epilogue_trunk_vs_unfused(dev):
    """The BN-folded trunk with FusedBasicBlock (bias-free MIOpen convolutions + hdn_bias_relu_f32) against the same folded
    trunk on PyTorch's own bias / add / relu kernels, NCHW and NHWC: the only arithmetic difference is (b2 + b_downsample)
    being added once."""
    from hdn_amd.trunk import fold_for_inference, resnet34_homo, FusedBasicBlock
    torch.manual_seed(3)
    net = resnet34_homo().eval().to(dev)
    gen = torch.Generator().manual_seed(4)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.empty(m.num_features).uniform_(-0.2, 0.2, generator=gen))
            m.running_var.copy_(torch.empty(m.num_features).uniform_(0.8, 1.2, generator=gen))
    x = torch.randn(8, 2, 127, 127, generator=gen).to(dev)
    with torch.no_grad():
        for cl in (False, True):
            plain = fold_for_inference(net, channels_last=cl, fused_stem=False, fused_epilogue=False)
            xin = x.contiguous(memory_format=torch.channels_last) if cl else x
            a = plain(xin)
            scale = float(a.abs().max())
            for mc in ((False, True) if cl else (False,)):
                fused = fold_for_inference(net, channels_last=cl, fused_stem=False, fused_epilogue=True, matrix_core=mc)
                assert sum(isinstance(m, FusedBasicBlock) for m in fused.modules()) == 16
                assert sum(m.p1 is not None for m in fused.modules() if isinstance(m, FusedBasicBlock)) == (13 if mc else 0)
                b = fused(xin)
                # MIOpen path: only (b2 + b_downsample) differs; matrix-core path: split-bf16 products, ~1e-6 relative per layer
                assert float((a - b).abs().max()) <= (1e-4 if mc else 2e-5) * scale, (cl, mc, float((a - b).abs().max()), scale)


def test_benchmarked_full_head_configuration_parity(dev):
    """The configuration bench.py's `full_head` block times — built by bench.build_full_head itself: B = 64, MIOpen find mode
    (cudnn.benchmark), BN-folded NHWC trunk, fused first stage, fused block epilogues — against the CPU oracle on a
    16-pair sample of the same batch: predicted corner offsets x within the north-star bound 1e-4, on two runs (find mode may pick
    split-K convolution kernels that accumulate atomically: the run-to-run difference is reported and bounded too)."""
    import bench
    from hdn_amd.homo_model import homo_stages
    g = torch.Generator().manual_seed(bench.SEED)
    imgs = torch.randn(bench.PAIRS, 2, 127, 127, generator=g)
    h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(bench.PAIRS, 1)
    saved = torch.backends.cudnn.benchmark
    try:
        net, data, cpu_sd = bench.build_full_head(dev, imgs.to(dev), h4p.to(dev))
        from hdn_amd.trunk import FusedBasicBlock, FusedStem
        fast = net._hdn_fast_trunk
        assert isinstance(fast.conv1, FusedStem) and sum(isinstance(m, FusedBasicBlock) for m in fast.modules()) == 16
        for _ in range(3):                     # the find-mode search happens on the first calls
            homo_stages(net, data)
        x1 = homo_stages(net, data)["x"].cpu()
        x2 = homo_stages(net, data)["x"].cpu()
    finally:
        torch.backends.cudnn.benchmark = saved
    idx = torch.arange(0, bench.PAIRS, 4)      # 16 of the 64 pairs
    ref_net = hdn_amd.HomoModelBuilder().eval()
    ref_net.load_state_dict(cpu_sd)
    sd = {k: v.clone() for k, v in ref_net.ShareFeature.state_dict().items()}
    sub = {"org_imgs": imgs[idx], "input_tensors": imgs[idx], "h4p": h4p[idx]}
    with torch.no_grad():
        _, _, _, aux = O.track_proj(sub, sd, lambda f: ref_net.fc(ref_net.avgpool(ref_net.backbone(f)).flatten(1)))
    e1, e2 = float((x1[idx] - aux["x"]).abs().max()), float((x2[idx] - aux["x"]).abs().max())
    rr = float((x1 - x2).abs().max())
    print(f"benchmarked full head: max|x - oracle| = {e1:.2e} / {e2:.2e} over 16 pairs, run-to-run max|dx| = {rr:.2e} over 64 pairs")
    assert e1 <= 1e-4 and e2 <= 1e-4, (e1, e2)
    assert rr <= 2e-5, rr


@pytest.mark.parametrize("B,C,S,O,nhwc", [(64, 512, 4, 8, True), (1, 512, 4, 8, True), (3, 512, 4, 8, False), (5, 96, 3, 16, True), (2, 700, 1, 1, False)])
def test_avgpool_fc_fused_tail(dev, B, C, S, O, nhwc):
    """hdn_avgpool_fc_f32 = fc(avgpool(x).flatten(1)) (homo_model_builder.py:161-165) against PyTorch-CPU in float64 and float32."""
    from hdn_amd.homo_model import avgpool_fc
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = torch.randn(B, C, S, S, generator=g).relu_()
    fc = torch.nn.Linear(C, O)
    fc.weight.data = torch.randn(O, C, generator=g) * 0.05
    fc.bias.data = torch.randn(O, generator=g)
    ref64 = torch.nn.functional.linear(x.double().mean(dim=(2, 3)), fc.weight.double(), fc.bias.double())
    ref32 = fc(torch.nn.functional.adaptive_avg_pool2d(x, 1).flatten(1))
    xd = x.to(dev)
    if nhwc:
        xd = xd.contiguous(memory_format=torch.channels_last)
    got = avgpool_fc(xd, fc.to(dev)).cpu().double()
    assert got.shape == (B, O)
    e_ref = float((ref32.double() - ref64).abs().max())
    assert float((got - ref64).abs().max()) <= 4 * e_ref + 1e-6, (float((got - ref64).abs().max()), e_ref)


@pytest.mark.parametrize("H,P,n,oc,ol", [(256, 625, 3, 2, 2), (256, 169, 3, 2, 4), (256, 961, 3, 2, 2), (128, 7, 2, 1, 8), (256, 33, 1, 2, 4)])
def test_head_tail_one_launch_vs_float64(dev, H, P, n, oc, ol):
    """hdn_head_tail_f32 (the first 1x1 convolution + folded BatchNorm + ReLU of the 2n (level, branch) heads on the matrix cores, then
    the second 1x1 convolution / loc_scale / weighted level sum as one folded product; ban.py:60-66,113-127) against the same two
    products in float64, held to PyTorch's own fp32 error on them."""
    from hdn_amd import heads as HD
    g = torch.Generator().manual_seed(H + P)
    Ho = P                          # (any [Ho, Wo] with Ho * Wo = P: the kernel sees pixels)
    feats = torch.randn(2 * n, H, Ho, 1, generator=g).relu_() * 3.0
    pk = HD._PackedHead()
    pk.w1 = (torch.randn(2 * n, H, H, generator=g) * 0.06).to(dev)
    pk.b1 = torch.randn(2 * n, H, 1, generator=g).to(dev)
    om = max(oc, ol)
    pk.wf = (torch.randn(2, om, n * H, generator=g) * 0.05).to(dev)
    pk.wf[0, oc:] = 0
    pk.wf[1, ol:] = 0
    pk.bf = torch.randn(2, om, 1, generator=g).to(dev)
    pk.w1p = HD._pack_w1(pk.w1)
    assert pk.w1p.dtype == torch.int16 and pk.w1p.numel() == 2 * (2 * n) * H * H
    got = HD.head_tail(feats.to(dev), pk, n).cpu().double()
    f64 = lambda t: t.detach().cpu().double()
    hid = (torch.baddbmm(f64(pk.b1), f64(pk.w1), f64(feats).view(2 * n, H, -1))).relu()
    ref = torch.baddbmm(f64(pk.bf), f64(pk.wf), hid.view(2, n * H, -1))
    hid32 = torch.baddbmm(pk.b1.cpu(), pk.w1.cpu(), feats.view(2 * n, H, -1)).relu()
    ref32 = torch.baddbmm(pk.bf.cpu(), pk.wf.cpu(), hid32.view(2, n * H, -1)).double()
    assert got.shape == ref.shape == (2, om, P)
    e_ref = float((ref32 - ref).abs().max())
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 4 * e_ref + 1e-6 * scale, (float((got - ref).abs().max()), e_ref, scale)
    # deterministic
    assert torch.equal(HD.head_tail(feats.to(dev), pk, n).cpu().double(), got)


@pytest.mark.parametrize("Hi,Wi,n,CO,nhwc", [(31, 31, 3, 512, False), (31, 31, 3, 512, True), (15, 15, 3, 512, True), (37, 37, 3, 512, False),
                                              (9, 14, 1, 64, True), (3, 3, 2, 32, False), (33, 20, 4, 96, False)])
def test_head_conv_search_one_launch_vs_float64(dev, Hi, Wi, n, CO, nhwc):
    """hdn_head_conv3x3_f32 (conv_search of the heads at B = 1: 3x3 / no padding, 256 input channels, folded bias + ReLU, NCHW planes out,
    n levels per launch; ban.py:55-59,75) against float64, held to PyTorch's own fp32 error on the same convolution."""
    from hdn_amd import heads as HD
    g = torch.Generator().manual_seed(Hi * 100 + Wi + CO)
    xs = [torch.randn(1, 256, Hi, Wi, generator=g).relu_() * 2.0 for _ in range(n)]
    ws = [torch.randn(CO, 256, 3, 3, generator=g) * 0.03 for _ in range(n)]
    bs = [torch.randn(CO, generator=g) for _ in range(n)]
    pk = HD._PackedHead()
    pk.wsp = HD._pack_conv_search([w.to(dev) for w in ws])
    pk.bsp = torch.stack(bs).to(dev)
    xd = [x.to(dev).contiguous(memory_format=torch.channels_last) if nhwc else x.to(dev) for x in xs]
    got = HD.head_conv_search(xd, pk).cpu().double()
    assert got.shape == (n, CO, Hi - 2, Wi - 2)
    for i in range(n):
        ref = torch.nn.functional.conv2d(xs[i].double(), ws[i].double(), bs[i].double()).relu()[0]
        ref32 = torch.nn.functional.conv2d(xs[i], ws[i], bs[i]).relu()[0].double()
        e_ref, scale = float((ref32 - ref).abs().max()), float(ref.abs().max())
        err = float((got[i] - ref).abs().max())
        assert err <= 4 * e_ref + 1e-6 * scale, (i, err, e_ref, scale)
    assert torch.equal(HD.head_conv_search(xd, pk).cpu().double(), got)          # deterministic


# --------------------------------------------------------------------------- the fp16-piece kernels' input range
def test_fp16_piece_range_guard(dev):
    """The two-fp16-piece kernels split their ACTIVATIONS as x 2^-8 (csrc/mfma_split.h), so with the guard OFF — the default — inputs far
    beyond fp16's 65,504 give finite results inside the float64 bound, as the reference's fp32 convolutions do (backbone/resnet.py:78-94,
    hdn/models/head/ban.py:55-66): checked at 7e4 (round-5 verdict) and 1e7, for every matrix-core entry point, on the outputs the large
    element reaches AND on the others (each set against its own scale).  The format ends at 65,520 x 256 = 1.67e7: beyond it (and for
    NaN) the guard (hdn_set_check_range / HDN_CHECK_RANGE=1; on in the -m gpu suite, tests/conftest.py) refuses the call with HDN_E_LIMIT."""
    import torch.nn.functional as F
    from hdn_amd import _lib, heads as HD
    from hdn_amd.trunk import pack_conv3x3, pack_conv3x3_v2, pack_conv3x3s2_ds, conv3x3_bias_relu, conv3x3s2_ds, FusedStem
    lib = _lib.load()
    prev = lib.hdn_set_check_range(0)
    cl = torch.channels_last

    def held(y, t64, ref32, reached, what):
        """y (device result), the float64 truth, the fp32 reference; `reached`: bool mask of outputs whose sum contains the large element."""
        y, t64, ref32 = y.detach().cpu().double(), t64.double(), ref32.double()
        assert torch.isfinite(y).all(), what
        for name, m in (("reached", reached), ("others", ~reached)):
            if m.any():
                e_ref, scale = float((ref32[m] - t64[m]).abs().max()), float(t64[m].abs().max())
                err = float((y[m] - t64[m]).abs().max())
                assert err <= 4 * e_ref + 1e-5 * scale, (what, name, err, e_ref, scale)

    try:
        g = torch.Generator().manual_seed(5)
        C, S, B = 256, 8, 24
        w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
        b = torch.randn(C, generator=g) * 0.1
        wp, wp2, bd = pack_conv3x3(w).to(dev), pack_conv3x3_v2(w).to(dev), b.to(dev)
        x = torch.randn(B, C, S, S, generator=g).clamp_min_(0)
        w2 = torch.randn(128, 64, 3, 3, generator=g) * 0.05
        wd = torch.randn(128, 64, 1, 1, generator=g) * 0.1
        wpd = pack_conv3x3s2_ds(w2, wd).to(dev)
        conv1 = torch.nn.Conv2d(2, 64, 7, 2, 3)
        st = FusedStem(conv1, True).to(dev)
        pk = HD._PackedHead()
        ws = torch.randn(64, 256, 3, 3, generator=g) * 0.03
        pk.wsp = HD._pack_conv_search([ws.to(dev)])
        pk.bsp = torch.zeros(1, 64, device=dev)
        H, P, n = 256, 169, 3
        pt = HD._PackedHead()
        pt.w1 = (torch.randn(2 * n, H, H, generator=g) * 0.06).to(dev)
        pt.b1 = torch.zeros(2 * n, H, 1, device=dev)
        pt.wf = (torch.randn(2, 4, n * H, generator=g) * 0.05).to(dev)
        pt.bf = torch.zeros(2, 4, 1, device=dev)
        pt.w1p = HD._pack_w1(pt.w1)
        f64 = lambda t: t.detach().cpu().double()
        for big in (7.0e4, -1.0e7):
            # 3x3 stride 1, both forms (the large element: sample B - 1, pixel (3, 5))
            xb = x.clone()
            xb[B - 1, 17, 3, 5] = abs(big)
            xd = xb.to(dev).contiguous(memory_format=cl)
            t = torch.relu(F.conv2d(xb.double(), w.double(), b.double(), padding=1))
            ref = torch.relu(F.conv2d(xb, w, b, padding=1))
            reached = torch.zeros_like(t, dtype=torch.bool)
            reached[B - 1, :, 2:5, 4:7] = True
            for kw in ({}, {"wpacked_v2": wp2}):
                held(conv3x3_bias_relu(xd, wp, bd, **kw), t, ref, reached, f"conv3x3 {big} {list(kw)}")
            # stride 2 + downsample
            x2 = torch.randn(3, 64, 32, 32, generator=g)
            x2[1, 5, 7, 9] = big
            o1, o2 = conv3x3s2_ds(x2.to(dev).contiguous(memory_format=cl), wpd, torch.zeros(128, device=dev))
            reach2 = torch.zeros((3, 128, 16, 16), dtype=torch.bool)
            reach2[1, :, 3:5, 4:6] = True                                 # input pixel (7, 9) under a 3 x 3 / stride 2 / pad 1 window
            held(o1, torch.relu(F.conv2d(x2.double(), w2.double(), None, stride=2, padding=1)), torch.relu(F.conv2d(x2, w2, None, stride=2, padding=1)),
                 reach2, f"conv3x3s2 {big}")
            held(o2, F.conv2d(x2.double(), wd.double(), None, stride=2), F.conv2d(x2, wd, None, stride=2), torch.zeros_like(reach2), f"downsample {big}")
            # the trunk's first stage on the matrix cores (conv 7x7 / 2 + ReLU + max-pool 3 / 2): the large element at the last pixel
            xs0 = torch.randn(8, 2, 127, 127, generator=g)
            xs0[7, 1, 126, 126] = big
            ys = st(xs0.to(dev))
            with torch.no_grad():
                ts = F.max_pool2d(torch.relu(F.conv2d(xs0.double(), conv1.weight.double(), conv1.bias.double(), stride=2, padding=3)), 3, 2, 1)
                rs = F.max_pool2d(torch.relu(conv1(xs0)), 3, 2, 1)
            reach_s = torch.zeros_like(ts, dtype=torch.bool)
            reach_s[7, :, -2:, -2:] = True
            held(ys, ts, rs, reach_s, f"stem {big}")
            # the heads' two kernels
            xs = torch.randn(1, 256, 9, 14, generator=g)
            xs[0, 200, 8, 13] = big
            yh = HD.head_conv_search([xs.to(dev)], pk)
            th, rh = torch.relu(F.conv2d(xs.double(), ws.double())), torch.relu(F.conv2d(xs, ws))
            reach_h = torch.zeros_like(th, dtype=torch.bool)
            reach_h[0, :, -1, -1] = True
            held(yh.reshape(th.shape), th, rh, reach_h, f"head conv_search {big}")
            feats = torch.randn(2 * n, H, P, 1, generator=g).relu_()
            feats[5, 255, 168, 0] = abs(big)
            yt = HD.head_tail(feats.to(dev), pt, n)
            hid = torch.baddbmm(f64(pt.b1), f64(pt.w1), f64(feats).view(2 * n, H, -1)).relu()
            tt = torch.baddbmm(f64(pt.bf), f64(pt.wf), hid.view(2, n * H, -1))
            hid32 = torch.baddbmm(pt.b1.cpu(), pt.w1.cpu(), feats.view(2 * n, H, -1)).relu()
            rt = torch.baddbmm(pt.bf.cpu(), pt.wf.cpu(), hid32.view(2, n * H, -1))
            reach_t = torch.zeros_like(tt, dtype=torch.bool)
            reach_t[:, :, 168] = True
            held(yt.reshape(tt.shape), tt, rt, reach_t, f"head tail {big}")
        # the guard: beyond 1.67e7 (and NaN) every entry point refuses, nothing is launched
        lib.hdn_set_check_range(1)
        for bad in (2.0e7, float("nan")):
            xb = x.clone()
            xb[B - 1, 17, 3, 5] = bad
            xd = xb.to(dev).contiguous(memory_format=cl)
            for kw in ({}, {"wpacked_v2": wp2}):
                with pytest.raises(ValueError, match="HDN_E_LIMIT"):
                    conv3x3_bias_relu(xd, wp, bd, **kw)
        x2 = torch.randn(3, 64, 32, 32, generator=g)
        x2[1, 5, 7, 9] = -2.0e7                                          # (the magnitude counts)
        with pytest.raises(ValueError, match="HDN_E_LIMIT"):
            conv3x3s2_ds(x2.to(dev).contiguous(memory_format=cl), wpd, torch.zeros(128, device=dev))
        xs0 = torch.randn(8, 2, 127, 127, generator=g)
        xs0[7, 1, 126, 126] = -2.0e7
        with pytest.raises(ValueError, match="HDN_E_LIMIT"):
            st(xs0.to(dev))
        xs = torch.randn(1, 256, 9, 14, generator=g)
        xs[0, 200, 8, 13] = 2.0e7
        with pytest.raises(ValueError, match="HDN_E_LIMIT"):
            HD.head_conv_search([xs.to(dev)], pk)
        feats = torch.randn(2 * n, H, P, 1, generator=g).relu_()
        feats[5, 255, 168, 0] = 2.0e7
        with pytest.raises(ValueError, match="HDN_E_LIMIT"):
            HD.head_tail(feats.to(dev), pt, n)
        # 1.6e7 is still inside
        xs[0, 200, 8, 13] = 1.6e7
        assert torch.isfinite(HD.head_conv_search([xs.to(dev)], pk)).all()
    finally:
        lib.hdn_set_check_range(prev)


# --------------------------------------------------------------------------- similarity backbone + necks: BatchNorm folded, epilogues fused
@pytest.mark.parametrize("nhwc", [False, True])
def test_backbone_folding_on_the_device(dev, nhwc):
    """hdn_amd.backbone.optimize_similarity_model on the production-shaped stand-in (the reference's module layout: tests/production_standin.py):
    backbone (ResNet-50, stride 8, dilated) and both necks through MIOpen convolutions + hdn_bias_relu_f32 passes against the same modules'
    own forward (BatchNorm / ReLU / add launches), 127- and 255-px crops; the switch is reversible and leaves state_dict alone."""
    import types
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import production_standin as PS
    from hdn_amd import backbone as BB
    torch.manual_seed(2)
    model = types.SimpleNamespace(backbone=PS.AtrousResNet50(), neck=PS.Necks(True), neck_lp=PS.Necks(False))
    for i, part in enumerate((model.backbone, model.neck, model.neck_lp)):
        PS._seed(part, 40 + i)
        for m in part.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.8, 1.3); m.bias.data.uniform_(-0.1, 0.1)
        part.to(dev).eval()
        if nhwc:
            part.to(memory_format=torch.channels_last)
    keys = list(model.backbone.state_dict().keys())
    xs = [torch.randn(1, 3, s, s, device=dev) * 60 + 110 for s in (127, 255)]
    if nhwc:
        xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    with torch.no_grad():
        ref = [(f, model.neck(f), model.neck_lp(f)) for f in (model.backbone(x) for x in xs)]
        assert BB.optimize_similarity_model(model, strict=True) == ["backbone", "neck", "neck_lp"]
        assert list(model.backbone.state_dict().keys()) == keys and type(model.backbone).__name__ == "AtrousResNet50"
        for x, (rf, rn, rl) in zip(xs, ref):
            f = model.backbone(x)
            for got, want in list(zip(f, rf)) + list(zip(model.neck(f), rn)) + list(zip(model.neck_lp(f), rl)):
                assert got.shape == want.shape
                assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()), (float((got - want).abs().max()), float(want.abs().max()))
        model.backbone.train()
        assert not BB._use_fused(model.backbone, xs[0])          # training mode: the class's own forward (BatchNorm statistics are live)
        model.backbone.eval()
        BB.restore_similarity_model(model)
        again = model.backbone(xs[0])
        assert "_hdn_fused" not in vars(model.backbone) and type(model.backbone) is PS.AtrousResNet50
        # (two runs of MIOpen's convolutions are not bit-identical: a tolerance, not torch.equal)
        assert all(float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) for a, b in zip(again, ref[0][0]))


def test_north_launch_events_are_one_shot_and_time_the_kernel(dev):
    """hdn_xcorr_north_launch_events: the next 31x31 (x) 61x61 launch carries the caller's hipEvent pair (hipExtLaunchKernelGGL) — elapsed time = the kernel's own
    duration; the hook is one-shot, and a correlation call of another shape in between DISARMS it (the events are never touched afterwards)."""
    import ctypes
    from hdn_amd import _lib
    hip = ctypes.CDLL("libamdhip64.so")
    lib = _lib.load()

    def ev():
        h = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(h)) == 0
        return h

    def elapsed(a, b):
        ms = ctypes.c_float(-1.0)
        return hip.hipEventElapsedTime(ctypes.byref(ms), a, b), ms.value

    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, 256, 61, 61, generator=g).clamp_min_(0).to(dev)
    k = torch.randn(8, 256, 31, 31, generator=g).clamp_min_(0).to(dev)
    xs = torch.randn(2, 256, 29, 29, generator=g).to(dev)
    ks = torch.randn(2, 256, 5, 5, generator=g).to(dev)
    ref = X.xcorr_depthwise(x, k)
    assert X.last_variant() == "north_fftc_61x61_31x31"
    torch.cuda.synchronize()
    # armed -> the launch records them; the result is the same launch's
    e0, e1 = ev(), ev()
    assert lib.hdn_xcorr_north_launch_events(e0, e1) == 0
    y = X.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    rc, ms = elapsed(e0, e1)
    assert rc == 0 and 0.003 < ms < 5.0, (rc, ms)          # 2,048 planes: ~15 us of kernel, not a host-side interval
    assert torch.equal(y, ref)
    # one-shot: the next launch does not re-record them
    X.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    assert elapsed(e0, e1) == (rc, ms)
    # armed, then a call of ANOTHER shape: disarmed, the following 31x31 launch leaves the pair untouched (never recorded -> an error from hipEventElapsedTime)
    f0, f1 = ev(), ev()
    assert lib.hdn_xcorr_north_launch_events(f0, f1) == 0
    X.xcorr_depthwise(xs, ks)
    X.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    assert elapsed(f0, f1)[0] != 0
    hip.hipGetLastError()          # (the expected failure above is also HIP's "last error": clear it, or the next launch check of this process reports it)
    for h in (e0, e1, f0, f1):
        hip.hipEventDestroy(h)
    assert hip.hipGetLastError() == 0
    X.xcorr_depthwise(xs, ks)      # and the library's own launch check is clean again
    torch.cuda.synchronize()
