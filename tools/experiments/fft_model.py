"""numpy model of xcorr_north_fft_kernel's algorithm (structure check; float32 arithmetic, not bit-exact to the GPU).

One wave = one pair of planes (A, B).  See DESIGN.md "north-star kernel, FFT form".
"""
import numpy as np

N = 64


def bitrev(p, bits):
    r = 0
    for i in range(bits):
        r |= ((p >> i) & 1) << (bits - 1 - i)
    return r


def dit_fft(a, sign, n=N):
    """Radix-2 DIT, a in natural order (complex64 [.., n]); sign=-1 forward, +1 inverse (unnormalised)."""
    bits = n.bit_length() - 1
    v = np.empty_like(a)
    for p in range(n):
        v[..., p] = a[..., bitrev(p, bits)]
    h = 1
    while h < n:
        for i in range(0, n, 2 * h):
            for j in range(h):
                w = np.complex64(np.exp(sign * 2j * np.pi * j / (2 * h)))
                x, y = v[..., i + j].copy(), v[..., i + j + h].copy()
                A = (x + w * y).astype(np.complex64)
                v[..., i + j] = A
                v[..., i + j + h] = (2 * x - A).astype(np.complex64)
        h *= 2
    return v


def half_fft_pruned(a32, pre_even, pre_odd, sign):
    """64 outputs of a sequence whose entries >= 32 are zero: even bins = FFT32(a*pre_even), odd bins = FFT32(a*pre_odd)."""
    out = np.zeros(a32.shape[:-1] + (N,), np.complex64)
    out[..., 0::2] = dit_fft((a32 * pre_even).astype(np.complex64), sign, 32)
    out[..., 1::2] = dit_fft((a32 * pre_odd).astype(np.complex64), sign, 32)
    return out


def corr_pair(xA, xB, kA, kB):
    """xA,xB [61,61], kA,kB [31,31] float32 -> outA,outB [31,31]."""
    j = np.arange(N)
    tau = np.exp(-1j * np.pi * j / N).astype(np.complex64)      # half-bin shift along a row
    w64 = np.exp(-2j * np.pi * j / N).astype(np.complex64)

    # x: row pass (lane r = row r), 61 active lanes
    c = np.zeros((N, N), np.complex64)
    c[:61, :61] = xA + 1j * xB
    C = dit_fft((c * tau).astype(np.complex64), -1)
    T = np.zeros((N, N), np.complex64)
    f = np.arange(32)
    T[:, f] = C[:, f] + np.conj(C[:, 63 - f])                     # 2 A(f+1/2)
    T[:, 32 + f] = -1j * (C[:, f] - np.conj(C[:, 63 - f]))        # 2 B(f+1/2)
    # x: column pass (lane c = column c), rows 61..63 are zero
    X = dit_fft(T.T.copy(), -1).T                                  # X[f1, c]

    # k: row pass, 31 active lanes, inputs 0..30 (pruned halves with the half-shift folded in)
    ck = np.zeros((32, 32), np.complex64)
    ck[:31, :31] = kA + 1j * kB
    Ck = half_fft_pruned(ck, tau[:32], (tau[:32] * w64[:32]).astype(np.complex64), -1)   # [32 rows, 64]
    Tk = np.zeros((32, N), np.complex64)
    Tk[:, f] = Ck[:, f] + np.conj(Ck[:, 63 - f])
    Tk[:, 32 + f] = -1j * (Ck[:, f] - np.conj(Ck[:, 63 - f]))
    # k: column pass, pruned input (rows 0..30), in two halves
    K = half_fft_pruned(Tk.T.copy(), np.ones(32, np.complex64), w64[:32], -1).T    # K[f1, c]

    R = (X * np.conj(K)).astype(np.complex64)
    # inverse column pass, rows 0..30 needed
    Y = dit_fft(R.T.copy(), +1).T[:31]                             # Y[r, c]
    # inverse row pass: Hermitian extension, 31 active lanes
    Cp = np.zeros((31, N), np.complex64)
    Cp[:, f] = Y[:, f] + 1j * Y[:, 32 + f]
    Cp[:, 63 - f] = np.conj(Y[:, f]) + 1j * np.conj(Y[:, 32 + f])
    cp = dit_fft(Cp, +1)[:, :31]
    post = (np.exp(1j * np.pi * j[:31] / N) / (4096.0 * 4.0)).astype(np.complex64)
    o = (cp * post).astype(np.complex64)
    return o.real.copy(), o.imag.copy()


if __name__ == "__main__":
    import torch
    rng = np.random.default_rng(1)
    xs = np.maximum(rng.standard_normal((2, 61, 61)), 0).astype(np.float32)
    ks = np.maximum(rng.standard_normal((2, 31, 31)), 0).astype(np.float32)
    oA, oB = corr_pair(xs[0], xs[1], ks[0], ks[1])
    truth = torch.nn.functional.conv2d(torch.from_numpy(xs.astype(np.float64))[None],
                                       torch.from_numpy(ks.astype(np.float64))[:, None], groups=2)[0].numpy()
    ref = torch.nn.functional.conv2d(torch.from_numpy(xs)[None], torch.from_numpy(ks)[:, None], groups=2)[0].numpy()
    got = np.stack([oA, oB])
    print("model vs f64: rms %.3g max %.3g | torch fp32 vs f64: rms %.3g max %.3g" % (
        np.sqrt(((got - truth) ** 2).mean()), np.abs(got - truth).max(),
        np.sqrt(((ref - truth) ** 2).mean()), np.abs(ref - truth).max()))
