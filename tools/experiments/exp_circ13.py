"""Timing of the 6-problem circular 13x13 launch at B=64."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
xs = [torch.randn(64, 256, 13, 13, generator=g).clamp_min_(0).to(dev) for _ in range(6)]
ks = [torch.randn(64, 256, 13, 13, generator=g).clamp_min_(0).to(dev) for _ in range(6)]
for _ in range(200): X.xcorr_depthwise_multi(xs, ks, circular=True)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): X.xcorr_depthwise_multi(xs, ks, circular=True)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 10)
print(X.last_variant(), " ".join(f"{t:.1f}" for t in ts), "us")
