"""5x5 (x) 29x29 / 35x35 rows-in-registers kernel: parity vs the oracle, time for the 6-problem launches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import hdn_amd
from hdn_amd import xcorr as X
from oracle import hdn_oracle as O
dev = torch.device("cuda:0")
r = np.random.default_rng(1)
for hx in (29, 35):
    for (B, C) in ((1, 1), (1, 3), (2, 9), (1, 37), (3, 50)):
        x = r.standard_normal((B, C, hx, hx), dtype=np.float32); k = r.standard_normal((B, C, 5, 5), dtype=np.float32)
        y = hdn_amd.xcorr_depthwise(torch.from_numpy(x).to(dev), torch.from_numpy(k).to(dev)).cpu().numpy()
        t = O.xcorr_depthwise_f64(x, k)
        print(hx, (B, C), X.last_variant(), "hip err %.3g" % np.abs(y - t).max())
for hx, Bn in ((29, 64), (35, 256)):
    n = 6 if hx == 29 else 2
    xs = [torch.randn(Bn, 256, hx, hx, device=dev) for _ in range(n)]
    ks = [torch.randn(Bn, 256, 5, 5, device=dev) for _ in range(n)]
    for _ in range(5): X.xcorr_depthwise_multi(xs, ks)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): X.xcorr_depthwise_multi(xs, ks)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 20)
    byt = n * Bn * 256 * (hx * hx + 25 + (hx - 4) ** 2) * 4
    print("%dx%d, %d problems x %d x 256 planes: %.1f us = %.2f TB/s" % (hx, hx, n, Bn, best, byt / best / 1e6))
