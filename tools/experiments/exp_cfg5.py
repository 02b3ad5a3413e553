"""Timing of the 5x5 (x) 35x35 kernel: 6 problems in one launch at B = 64 (bench.py --config 5's correlation launch).
(Round 3's A/B runs against the persistent LDS-DMA form and the occupancy sweep: profiles/round3_cfg5_experiments.txt.)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B = 64
xs = [[torch.randn(B, 256, 35, 35, generator=g).clamp_min_(0).to(dev) for _ in range(6)] for _ in range(2)]
ks = [[torch.randn(B, 256, 5, 5, generator=g).clamp_min_(0).to(dev) for _ in range(6)] for _ in range(2)]
for _ in range(300): X.xcorr_depthwise_multi(xs[0], ks[0])
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): X.xcorr_depthwise_multi(xs[i & 1], ks[i & 1])
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1000 / 20)
nbytes = 6 * 4 * 256 * (1225 + 25 + 961) * B
print(f"{X.last_variant()}: " + " ".join(f"{t:.1f}" for t in ts) + f" us; {nbytes / min(ts) / 1e3:.0f} GB/s algorithmic = {nbytes / min(ts) / 8e6:.3f} of 8 TB/s")
