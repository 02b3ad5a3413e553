"""The fused kernel's LDS solve must give the bit-identical H of the standalone (shuffle) solve."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import homography as G
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
bad = 0
for scale in (0.0, 1.0, 8.0, 30.0, 100.0):
    B = 512
    src = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(B, 1) + (scale / 4) * torch.randn(B, 8, generator=g)
    off = scale * torch.randn(B, 8, generator=g)
    img = torch.randn(B, 1, 127, 127, generator=g)
    H1 = G.DLT_solve(src.to(dev), off.to(dev)).reshape(B, 9)
    H2, _ = G.dlt_warp(src.to(dev), off.to(dev), img.to(dev))
    d = (H1 != H2.reshape(B, 9)) & ~(torch.isnan(H1) & torch.isnan(H2.reshape(B, 9)))
    bad += int(d.sum())
    print(scale, "mismatching entries:", int(d.sum()), "max diff", float((H1 - H2.reshape(B, 9)).abs().nan_to_num().max()))
print("BITEXACT" if bad == 0 else "DIFFERENT")
