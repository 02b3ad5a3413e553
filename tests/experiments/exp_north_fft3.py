"""Correctness + timing of the two-waves-per-SIMD FFT kernel (variant fft2w) against fft and the float64 truth."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import hdn_amd
from hdn_amd import xcorr as X
from oracle import hdn_oracle as O
dev = torch.device("cuda:0")
g = np.random.default_rng(3)
for (B, C) in ((1, 4), (2, 8), (3, 12)):
    x = torch.from_numpy(np.maximum(g.standard_normal((B, C, 61, 61), dtype=np.float32), 0))
    k = torch.from_numpy(np.maximum(g.standard_normal((B, C, 31, 31), dtype=np.float32), 0))
    truth = torch.nn.functional.conv2d(x.double().reshape(1, B*C, 61, 61), k.double().reshape(B*C, 1, 31, 31), groups=B*C).reshape(B, C, 31, 31)
    for v in ("fft", "fft2w"):
        with X.north_variant(v):
            y = X.xcorr_depthwise(x.to(dev), k.to(dev)).cpu()
        print(B, C, v, "vs f64 rms %.3g max %.3g" % ((y - truth).pow(2).mean().sqrt(), (y - truth).abs().max()))
x = torch.relu(torch.randn(64, 256, 61, 61, device=dev)); k = torch.relu(torch.randn(64, 256, 31, 31, device=dev))
for v in ("fft", "fft2w", "fft", "fft2w"):
    with X.north_variant(v):
        for _ in range(300): y = X.xcorr_depthwise(x, k)   # clock ramp
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): y = X.xcorr_depthwise(x, k)
        e1.record(); torch.cuda.synchronize()
        print(v, "B=64 C=256: %.1f us per launch" % (e0.elapsed_time(e1) * 1000 / 50))
        b, c = 63, 255
        t = torch.nn.functional.conv2d(x[b, c].double().cpu()[None, None], k[b, c].double().cpu()[None, None])[0, 0]
        print("   last plane max err vs f64 %.3g" % (y[b, c].cpu() - t).abs().max())
