"""Where do the copy launches of a full-head step (B = 64) come from?  torch.profiler, CPU op -> device activity."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from hdn_amd.homo_model import homo_stages
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(bench.SEED)
imgs = torch.randn(64, 2, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(64, 1).to(dev)
net, data, _ = bench.build_full_head(dev, imgs, h4p)
for _ in range(10): homo_stages(net, data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    homo_stages(net, data); torch.cuda.synchronize()
evs = prof.events()
for e in evs:
    n = e.name
    if e.device_type == torch.autograd.DeviceType.CUDA and ("copy" in n.lower() or "Memcpy" in n or "elementwise" in n or "reduce" in n or "Cijk" in n):
        print("%-70s %7.1f us" % (n[:70], e.device_time if hasattr(e, "device_time") else e.cuda_time))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60, max_src_column_width=90))
