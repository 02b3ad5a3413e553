"""The three stride-2 stages at B = 64 through both kernels (hdn_conv3x3s2_ds_f32 / hdn_conv3x3s2_v2_f32), for rocprofv3 kernel times."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd.trunk import pack_conv3x3s2_ds, pack_conv3x3s2_ds_v2, conv3x3s2_ds
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "64"))
g = torch.Generator().manual_seed(1)
for CI, S in ((64, 16), (128, 8), (256, 4)):
    CO = 2 * CI
    w = torch.randn(CO, CI, 3, 3, generator=g) * (2.0 / (9 * CI)) ** 0.5
    wd = torch.randn(CO, CI, 1, 1, generator=g) * (1.0 / CI) ** 0.5
    b = (torch.randn(CO, generator=g) * 0.1).to(dev)
    x = torch.randn(B, CI, 2 * S, 2 * S, generator=g).clamp_min_(0).to(dev).contiguous(memory_format=torch.channels_last)
    wp, wp2 = pack_conv3x3s2_ds(w, wd).to(dev), pack_conv3x3s2_ds_v2(w, wd).to(dev)
    for _ in range(40):
        conv3x3s2_ds(x, wp, b)
        conv3x3s2_ds(x, wp, b, wpacked_v2=wp2)
torch.cuda.synchronize()
