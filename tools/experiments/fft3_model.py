"""numpy model of the two-waves-per-SIMD FFT layout (xcorr_north_fft3_kernel): one pair of planes per wave as before, but the
column passes run one PLANE at a time with 64 lanes = 32 half-bin columns x 2 halves of the 64 bins (DIF split with a
per-lane twiddle), so the LDS image is 61 x 32 complex (16 KB) instead of 61 x 64.  Structure check only (complex64).
"""
import numpy as np
from fft_model import dit_fft, half_fft_pruned, N

j = np.arange(N)
tau = np.exp(-1j * np.pi * j / N).astype(np.complex64)
w64 = np.exp(-2j * np.pi * j / N).astype(np.complex64)


def col_pass_dif(T, nrows, sign=-1):
    """T[r, c] (r < nrows <= 64 valid, rest zero): 64-pt FFT down the rows, returned as (even bins [32, C], odd bins [32, C]).
    lane (c, h): t_r = a_r + s_h a_{r+32}; u_r = t_r * tw_h(r); FFT32."""
    a = np.zeros((N, T.shape[1]), np.complex64)
    a[:nrows] = T[:nrows]
    tw = np.exp(sign * 2j * np.pi * np.arange(32) / N).astype(np.complex64)[:, None]
    ev = dit_fft((a[:32] + a[32:]).T.copy(), sign, 32).T
    od = dit_fft(((a[:32] - a[32:]) * tw).T.copy(), sign, 32).T
    return ev.astype(np.complex64), od.astype(np.complex64)


def corr_pair3(xA, xB, kA, kB):
    f = np.arange(32)
    # ---- x row pass (pair-packed, as v2): SA, SB' per row
    c = np.zeros((N, N), np.complex64)
    c[:61, :61] = xA + 1j * xB
    C = dit_fft((c * tau).astype(np.complex64), -1)
    SA = (C[:, f] + np.conj(C[:, 63 - f])).astype(np.complex64)      # [64 rows, 32]   2 A(f+1/2)
    SB = (C[:, f] - np.conj(C[:, 63 - f])).astype(np.complex64)      # 2 i B(f+1/2)
    # ---- column passes, one plane at a time; lane (c, h) ends with bins 2g+h
    XA = col_pass_dif(SA, 61)
    XB = col_pass_dif(SB, 61)
    # ---- k row pass (pair-packed, pruned), raw spectrum; split while the column lanes read
    ck = np.zeros((32, 32), np.complex64)
    ck[:31, :31] = kA + 1j * kB
    Ck = half_fft_pruned(ck, tau[:32], (tau[:32] * w64[:32]).astype(np.complex64), -1)      # [32 rows, 64]
    KAr = (Ck[:, f] + np.conj(Ck[:, 63 - f])).astype(np.complex64)
    KBr = (Ck[:, f] - np.conj(Ck[:, 63 - f])).astype(np.complex64)
    KA = col_pass_dif(KAr, 31)
    KB = col_pass_dif(KBr, 31)
    out = []
    Y = []
    for X, K in ((XA, KA), (XB, KB)):
        Re, Ro = (X[0] * np.conj(K[0])).astype(np.complex64), (X[1] * np.conj(K[1])).astype(np.complex64)   # bins 2g, 2g+1
        # inverse column pass, DIT: y[r] = E'[r] + w64^{-r} O'[r]; E' = IFFT32 over even bins (lane h=0), O' over odd (h=1)
        E = dit_fft(Re.T.copy(), +1, 32).T          # [32 (r), C]
        O = dit_fft(Ro.T.copy(), +1, 32).T
        r = np.arange(31)[:, None]
        Y.append((E[:31] + np.exp(2j * np.pi * r / N).astype(np.complex64) * O[:31]).astype(np.complex64))  # combine by the ROW lanes
    YA, YB = Y
    # ---- inverse row pass (pair-packed as v2).  Note SB' = 2iB: the two factors i cancel in X conj(K), so YB is plane B's.
    Cp = np.zeros((31, N), np.complex64)
    Cp[:, f] = YA + 1j * YB
    Cp[:, 63 - f] = np.conj(YA) + 1j * np.conj(YB)
    cp = dit_fft(Cp, +1)[:, :31]
    post = (np.exp(1j * np.pi * j[:31] / N) / (4096.0 * 4.0)).astype(np.complex64)
    o = (cp * post).astype(np.complex64)
    return o.real.copy(), o.imag.copy()


if __name__ == "__main__":
    import torch
    rng = np.random.default_rng(1)
    xs = np.maximum(rng.standard_normal((2, 61, 61)), 0).astype(np.float32)
    ks = np.maximum(rng.standard_normal((2, 31, 31)), 0).astype(np.float32)
    oA, oB = corr_pair3(xs[0], xs[1], ks[0], ks[1])
    truth = torch.nn.functional.conv2d(torch.from_numpy(xs.astype(np.float64))[None],
                                       torch.from_numpy(ks.astype(np.float64))[:, None], groups=2)[0].numpy()
    got = np.stack([oA, oB])
    print("model3 vs f64: rms %.3g max %.3g" % (np.sqrt(((got - truth) ** 2).mean()), np.abs(got - truth).max()))
