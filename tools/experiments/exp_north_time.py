"""Back-to-back and solo-bracket timing of one 31x31 (x) 61x61 variant at B = 64, C = 256 (long clock warm-up), plus its error against
float64 on the last plane.   usage: [HDN_LIB_PATH=...] exp_north_time.py <variant> [<variant> ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(64, 256, 61, 61, generator=g).clamp_min_(0).to(dev)
k = torch.randn(64, 256, 31, 31, generator=g).clamp_min_(0).to(dev)
t = torch.nn.functional.conv2d(x[63, 255].double().cpu()[None, None], k[63, 255].double().cpu()[None, None])[0, 0]
for v in sys.argv[1:]:
    with X.north_variant(v):
        for _ in range(400): y = X.xcorr_depthwise(x, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): y = X.xcorr_depthwise(x, k)
        e1.record(); torch.cuda.synchronize()
        b2b = e0.elapsed_time(e1) * 1000 / 50
        solo = []
        for _ in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); y = X.xcorr_depthwise(x, k); b.record(); torch.cuda.synchronize()
            solo.append(a.elapsed_time(b) * 1000)
        err = (y[63, 255].cpu() - t).abs().max().item()
        print(f"{os.path.basename(os.environ.get('HDN_LIB_PATH', 'libhdn_hip.so')):24s} {v:6s} [{X.last_variant()}]: back to back {b2b:.1f} us; "
              f"solo brackets min {min(solo):.1f} median {np.median(solo):.1f} us; last plane max err vs f64 {err:.2e}")
