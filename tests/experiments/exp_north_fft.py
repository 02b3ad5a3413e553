import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import hdn_amd
from hdn_amd import xcorr as X
from oracle import hdn_oracle as O
dev = torch.device("cuda:0")
g = np.random.default_rng(3)
for (B, C) in ((1, 2), (1, 3), (2, 8), (3, 7)):
    x = torch.from_numpy(np.maximum(g.standard_normal((B, C, 61, 61), dtype=np.float32), 0))
    k = torch.from_numpy(np.maximum(g.standard_normal((B, C, 31, 31), dtype=np.float32), 0))
    y = X.xcorr_depthwise(x.to(dev), k.to(dev)).cpu()
    ref = O.xcorr_depthwise(x, k)
    truth = torch.nn.functional.conv2d(x.double().reshape(1, B*C, 61, 61), k.double().reshape(B*C, 1, 31, 31), groups=B*C).reshape(B, C, 31, 31)
    print(B, C, X.last_variant(), "vs ref max %.3g | vs f64 rms %.3g max %.3g | ref vs f64 rms %.3g max %.3g" % (
        (y - ref).abs().max(), (y - truth).pow(2).mean().sqrt(), (y - truth).abs().max(), (ref - truth).pow(2).mean().sqrt(), (ref - truth).abs().max()))
# timing B=64
x = torch.relu(torch.randn(64, 256, 61, 61, device=dev)); k = torch.relu(torch.randn(64, 256, 31, 31, device=dev))
for _ in range(3): y = X.xcorr_depthwise(x, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y = X.xcorr_depthwise(x, k)
e1.record(); torch.cuda.synchronize()
print("B=64 C=256: %.1f us per launch" % (e0.elapsed_time(e1) * 1000 / 20), X.last_variant())
# check a few planes against f64
idx = [(0, 0), (13, 77), (63, 255), (63, 254)]
for (b, c) in idx:
    t = torch.nn.functional.conv2d(x[b, c].double().cpu()[None, None], k[b, c].double().cpu()[None, None])[0, 0]
    print("plane", b, c, "max err vs f64 %.3g" % (y[b, c].cpu() - t).abs().max())
