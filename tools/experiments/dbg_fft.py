import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from hdn_amd import xcorr as X
from oracle import hdn_oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
x = torch.randn(1, 4, 61, 61, generator=g); k = torch.zeros(1, 4, 31, 31)
# delta kernels: k[u0,v0] = 1 -> out[i,j] = x[i+u0, j+v0]
for (u0, v0) in ((0, 0), (0, 1), (1, 0), (0, 2), (5, 7)):
    k.zero_(); k[:, :, u0, v0] = 1.0
    y = X.xcorr_depthwise(x.to(dev), k.to(dev)).cpu()
    ref = O.xcorr_depthwise(x, k)
    e = (y - ref).abs()
    print((u0, v0), X.last_variant(), "max err per plane", [float(e[0, c].max()) for c in range(4)])
k = torch.randn(1, 4, 31, 31, generator=g)
y = X.xcorr_depthwise(x.to(dev), k.to(dev)).cpu(); ref = O.xcorr_depthwise(x, k)
print("random", float((y - ref).abs().max()), float(ref.abs().max()))
