import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from hdn_amd import _lib, xcorr as X
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
x = torch.randn(1, 4, 61, 61, generator=g); k = torch.randn(1, 4, 31, 31, generator=g)
y = X.xcorr_depthwise(x.to(dev), k.to(dev)); torch.cuda.synchronize()
lib = _lib.load(); lib.hdn_debug_read_kimg.argtypes = [ctypes.c_void_p]
buf = np.zeros(8192, np.float32); assert lib.hdn_debug_read_kimg(buf.ctypes.data) == 0
img = buf[:31 * 65 * 2].reshape(31, 65, 2); got = img[:, :64, 0] + 1j * img[:, :64, 1]
c = (k[0, 0].numpy() + 1j * k[0, 1].numpy()).astype(np.complex128)   # pair 0 = planes 0 (A), 1 (B)
j = np.arange(31); f = np.arange(64)
want = np.array([[(c[r] * np.exp(-2j * np.pi * j * (ff + 0.5) / 64)).sum() for ff in f] for r in range(31)])
err = np.abs(got - want)
print("max err even bins", err[:, 0::2].max(), "odd bins", err[:, 1::2].max())
print("per-row max err:", np.round(err.max(axis=1), 3))
print("row 0 odd bins got :", np.round(got[0, 1:9:2], 3)); print("row 0 odd bins want:", np.round(want[0, 1:9:2], 3))
for name, sl, mul in (("even", slice(0, 64, 2), 1), ("odd", slice(1, 64, 2), 3)):
    R = (got - want)[:, sl]
    r = np.fft.ifft(R, axis=1)            # residual per input index j (twiddled)
    print(name, "row 0 |residual input| per j:", np.round(np.abs(r[0]), 2))
    print(name, "row 5 |residual input| per j:", np.round(np.abs(r[5]), 2))
    jj = np.argmax(np.abs(r[0])); print("   row 0: j =", jj, "residual/ (c_j tw) =", r[0, jj] / (c[0, jj % 31] * np.exp(-1j * np.pi * mul * jj / 64)) if jj < 31 else None, "c_j =", c[0, jj % 31])
