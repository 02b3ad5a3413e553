"""13x13 circular correlation, DFT form: parity vs the float64 oracle, time for the 6-problem launch of 64 x 256 planes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import hdn_amd
from hdn_amd import xcorr as X
from oracle import hdn_oracle as O
dev = torch.device("cuda:0")
r = np.random.default_rng(1)
for (B, C) in ((1, 1), (1, 3), (2, 9), (1, 36), (1, 37), (3, 50), (2, 256)):
    x = r.standard_normal((B, C, 13, 13), dtype=np.float32); k = r.standard_normal((B, C, 13, 13), dtype=np.float32)
    y = hdn_amd.xcorr_depthwise_circular(torch.from_numpy(x).to(dev), torch.from_numpy(k).to(dev)).cpu().numpy()
    t = O.xcorr_depthwise_circular_f64(x, k)
    ref = O.xcorr_depthwise_circular(torch.from_numpy(x), torch.from_numpy(k)).numpy()
    print((B, C), X.last_variant(), "hip err %.3g  ref err %.3g" % (np.abs(y - t).max(), np.abs(ref - t).max()))
xs = [torch.randn(64, 256, 13, 13, device=dev) for _ in range(6)]
ks = [torch.randn(64, 256, 13, 13, device=dev) for _ in range(6)]
for _ in range(10): X.xcorr_depthwise_multi(xs, ks, circular=True)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): X.xcorr_depthwise_multi(xs, ks, circular=True)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1000 / 30)
print("6 x 64 x 256 planes: %.1f us" % best)
