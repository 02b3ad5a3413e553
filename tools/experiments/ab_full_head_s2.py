"""Same-box A/B of the full head (bench.build_full_head, B = 64): the three stride-2 stages through hdn_conv3x3s2_v2_f32 (round 5) against
hdn_conv3x3s2_ds_f32 (round 4), alternating in one process."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from hdn_amd.homo_model import homo_stages
from hdn_amd.trunk import FusedBasicBlock
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(bench.SEED)
imgs = torch.randn(64, 2, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(64, 1).to(dev)
net, data, _ = bench.build_full_head(dev, imgs, h4p)
def timed(n=60):
    for _ in range(10): homo_stages(net, data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): homo_stages(net, data)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
xs = {}
for rep in range(4):
    for off in (True, False):
        FusedBasicBlock.v2_s2_disabled = off
        ms = timed()
        xs.setdefault(off, []).append(ms)
        print("stride-2 stages on the round-%d kernel: %.4f ms per 64 pairs = %.1f k frames/s" % (4 if off else 5, ms, 64 / ms), flush=True)
FusedBasicBlock.v2_s2_disabled = True
a = homo_stages(net, data)["x"]
FusedBasicBlock.v2_s2_disabled = False
b = homo_stages(net, data)["x"]
print("best: round 4 %.4f ms, round 5 %.4f ms (%.1f %%); max |x4 - x5| = %.2e" % (min(xs[True]), min(xs[False]), 100 * (min(xs[True]) / min(xs[False]) - 1), float((a - b).abs().max())))
