"""Randomised differential test of the correlation entry points against the CPU oracle (run on the GPU box).

    python tests/tools/fuzz_parity.py [seconds]
Random plane counts / shapes / variants / pointer alignments / data scales; every result is held to the parity bound of
tests/test_gpu_parity.py::check_xcorr.  Prints a summary line; exits non-zero on the first violation.
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import hdn_amd
from hdn_amd import xcorr as X
from oracle import hdn_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(time.time()) if os.environ.get("FUZZ_SEED") is None else int(os.environ["FUZZ_SEED"]))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def shifted(a, off):
    flat = torch.zeros(a.size + off, device=dev)
    flat[off:] = T(a).to(dev).reshape(-1)
    return flat[off:].view(*a.shape)


def check(got, x, k, circular, tag):
    f64 = O.xcorr_depthwise_circular_f64 if circular else O.xcorr_depthwise_f64
    ref = (O.xcorr_depthwise_circular if circular else O.xcorr_depthwise)(T(x), T(k)).numpy()
    truth, mag = f64(x, k), f64(np.abs(x), np.abs(k))
    got = got.cpu().numpy()
    pm = mag
    if x.shape[-1] == 61:  # FFT variants: planes are transformed in pairs
        P = mag.reshape(-1, *mag.shape[2:])
        if P.shape[0] % 2 == 0:
            pmax = np.maximum(P[0::2], P[1::2]).max(axis=(1, 2), keepdims=True).repeat(2, axis=0)
            pm = np.maximum(P, pmax).reshape(mag.shape)
    bad = np.abs(got - ref) > 1e-4 + 2e-6 * pm
    assert not bad.any(), f"{tag}: max|hip-ref|={np.abs(got - ref).max():.3e} at {np.argwhere(bad)[0]}"
    e_hip, e_ref = np.abs(got - truth).max(), np.abs(ref - truth).max()
    assert e_hip <= 2 * e_ref + 1e-6 + 2e-6 * pm.max(), f"{tag}: hip err {e_hip:.3e} vs reference err {e_ref:.3e}"


shapes = [(61, 31, False), (29, 5, False), (35, 5, False), (13, 13, True), (None, None, False), (None, None, True)]
variants = ["fft", "direct"]
only_fft = os.environ.get("FUZZ_FFT") == "1"  # aligned 31x31 (x) 61x61 problems on the two FFT kernels only
if only_fft:
    shapes, variants = shapes[:1], variants[:3]
t0, n, seen = time.time(), 0, {}
while time.time() - t0 < budget:
    hx, hk, circ = shapes[rng.integers(len(shapes))]
    if hx is None:
        hx, wx = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        hk, wk = int(rng.integers(1, hx + 1)), int(rng.integers(1, wx + 1))
    else:
        wx, wk = hx, hk
    B, C = int(rng.integers(1, 5)), int(rng.integers(1, 70 if hx != 61 else 24))
    scale = float(10.0 ** rng.integers(-3, 3))
    signed = bool(rng.integers(2))
    x = rng.standard_normal((B, C, hx, wx), dtype=np.float32) * scale
    k = rng.standard_normal((B, C, hk, wk), dtype=np.float32)
    if not signed:
        x, k = np.maximum(x, 0), np.maximum(k, 0)
    ox, ok = int(rng.integers(0, 4)) * int(rng.integers(2)), int(rng.integers(0, 4)) * int(rng.integers(2))
    if only_fft:
        ox = ok = 0
        C = int(rng.integers(1, 40))
    v = variants[rng.integers(len(variants))]
    fn = X.xcorr_depthwise_circular if circ else X.xcorr_depthwise
    with X.north_variant(v):
        y = fn(shifted(x, ox), shifted(k, ok))
        tag = f"{(B, C, hx, wx, hk, wk)} circ={circ} variant={v} off=({ox},{ok}) scale={scale} signed={signed} -> {X.last_variant()}"
        seen[X.last_variant()] = seen.get(X.last_variant(), 0) + 1
    check(y, x, k, circ, tag)
    n += 1
print(f"fuzz: {n} cases in {time.time() - t0:.0f} s, all within bounds; kernels hit: {seen}")
