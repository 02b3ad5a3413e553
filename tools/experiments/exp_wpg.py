"""A/B: single-wave workgroups vs 4 (v2) / 8 (v3) autonomous waves per workgroup for the FFT kernels (HDN_FFT_WPG is read at
the first launch, so each arm is its own process).  usage: python tools/experiments/exp_wpg.py <variant>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
v = sys.argv[1]
g = torch.Generator().manual_seed(1)
x = torch.randn(64, 256, 61, 61, generator=g).clamp_min_(0).to(dev)
k = torch.randn(64, 256, 31, 31, generator=g).clamp_min_(0).to(dev)
with X.north_variant(v):
    for _ in range(400): X.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): X.xcorr_depthwise(x, k)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 10)
    y = X.xcorr_depthwise(x, k)
    print(f"{v} HDN_FFT_WPG={os.environ.get('HDN_FFT_WPG', 'default')} {X.last_variant()}: " + " ".join(f"{t:.1f}" for t in ts) + " us; checksum %.6e" % y.double().sum().item())
