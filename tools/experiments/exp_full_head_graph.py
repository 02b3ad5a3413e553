"""Full head at B = 64 (bench.build_full_head): eager launches vs one hipGraph replay per step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from hdn_amd.homo_model import homo_stages
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(bench.SEED)
imgs = torch.randn(64, 2, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(64, 1).to(dev)
net, data, _ = bench.build_full_head(dev, imgs, h4p)
for _ in range(10): homo_stages(net, data)
torch.cuda.synchronize()
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager  %.3f ms/step" % timed(lambda: homo_stages(net, data)))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): homo_stages(net, data)
torch.cuda.current_stream().wait_stream(side)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = homo_stages(net, data)
print("graph  %.3f ms/step" % timed(gr.replay))
