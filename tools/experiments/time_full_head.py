"""Full head (bench.build_full_head, B = 64) wall time per step, best of 5 x 60 steps: for same-box A/B of environment switches."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from hdn_amd.homo_model import homo_stages
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(bench.SEED)
B = int(os.environ.get("B", "64"))
imgs = torch.randn(B, 2, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(B, 1).to(dev)
net, data, _ = bench.build_full_head(dev, imgs, h4p)
for _ in range(20): homo_stages(net, data)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): homo_stages(net, data)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 60 * 1e3)
print("%s full head: %.4f ms per %d pairs = %.1f k frames/s" % (os.environ.get("TAG", ""), best, B, B / best), flush=True)
