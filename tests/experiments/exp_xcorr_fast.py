"""xcorr_fast (UPChannelBAN's channel-contracting correlation): parity vs the oracle on a few shapes, time at the production shape."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import hdn_amd
from oracle import hdn_oracle as O
dev = torch.device("cuda:0")
r = np.random.default_rng(0)
for (B, C, Oo, hx, hk) in ((2, 16, 2, 29, 5), (1, 7, 4, 12, 3), (3, 256, 4, 29, 5), (1, 5, 1, 9, 9)):
    x = r.standard_normal((B, C, hx, hx), dtype=np.float32); k = r.standard_normal((B, Oo * C, hk, hk), dtype=np.float32)
    y = hdn_amd.xcorr_fast(torch.from_numpy(x).to(dev), torch.from_numpy(k).to(dev)).cpu().numpy()
    ref = O.xcorr_fast(torch.from_numpy(x), torch.from_numpy(k)).numpy()
    print((B, C, Oo, hx, hk), "max|hip - ref| %.3g (|ref| max %.3g)" % (np.abs(y - ref).max(), np.abs(ref).max()))
for Oo in (2, 4):
    x = torch.randn(64, 256, 29, 29, device=dev); k = torch.randn(64, Oo * 256, 5, 5, device=dev)
    for _ in range(3): hdn_amd.xcorr_fast(x, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hdn_amd.xcorr_fast(x, k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("B=64 C=256 O=%d 29x29 (x) 5x5: %.1f us = %.1f TFLOP/s" % (Oo, us, 64 * Oo * 625 * 6400 * 2 / us / 1e6))
