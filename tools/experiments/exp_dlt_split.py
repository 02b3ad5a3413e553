"""Is the fused DLT+warp launch (every workgroup waits for its wave 0's serial fp64 solve) slower than solve + warp as two launches?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import hdn_amd
from hdn_amd import homography as G, share_feature as SF
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
B = 64
img = torch.randn(B, 1, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(B, 1).to(dev)
off = (8 * torch.randn(B, 8, generator=g)).to(dev)
M = torch.tensor([[63.5, 0, 63.5], [0, 63.5, 63.5], [0, 0, 1]], device=dev).expand(B, 3, 3)
Mi = torch.inverse(M[0]).expand(B, 3, 3)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
H = G.DLT_solve(h4p, off).squeeze(1)
th = (Mi @ H @ M).contiguous()
print("fused dlt_warp      %.1f us" % t(lambda: G.dlt_warp(h4p, off, img)))
print("DLT_solve alone     %.1f us" % t(lambda: G.DLT_solve(h4p, off)))
print("transformer alone   %.1f us" % t(lambda: G.transformer(img, th, (127, 127), want_condition=False)))
print("empty-ish launch    %.1f us" % t(lambda: G.l1_score(img[0, 0], img[1, 0], 1.0)))
