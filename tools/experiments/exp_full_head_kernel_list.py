"""Every device kernel of ONE full-head step (bench.build_full_head, B = 64) in launch order, with durations (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from hdn_amd.homo_model import homo_stages
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(bench.SEED)
imgs = torch.randn(64, 2, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(64, 1).to(dev)
net, data, _ = bench.build_full_head(dev, imgs, h4p)
for _ in range(20): homo_stages(net, data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    homo_stages(net, data)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
tot = 0.0
t0 = evs[0].time_range.start
for e in evs:
    d = e.time_range.end - e.time_range.start
    tot += d
    print("%8.1f us  +%6.1f  %s" % (e.time_range.start - t0, d, e.name[:110]))
print("kernels: %d, sum %.1f us, span %.1f us" % (len(evs), tot, evs[-1].time_range.end - t0))
