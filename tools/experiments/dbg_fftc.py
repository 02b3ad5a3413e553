import sys, numpy as np, torch
sys.path.insert(0, ".")
from hdn_amd import xcorr as X
from oracle import hdn_oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for (B, C) in ((1, 2), (1, 4), (2, 8), (1, 3), (5, 51), (1, 1)):
    x = torch.randn(B, C, 61, 61, generator=g).clamp_min_(0); k = torch.randn(B, C, 31, 31, generator=g).clamp_min_(0)
    with X.north_variant("fftc"):
        y = X.xcorr_depthwise(x.to(dev), k.to(dev)).cpu(); v = X.last_variant()
    ref = O.xcorr_depthwise(x, k)
    print((B, C), v, "max err", float((y - ref).abs().max()), "ref max", float(ref.abs().max()))
