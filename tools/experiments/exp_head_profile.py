"""Device time per kernel of one packed MultiBAN / MultiCircBAN forward at B = 1, 256 channels (torch.profiler): what the library convolutions cost."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from torch.profiler import profile, ProfilerActivity
from conftest import seeded_head256
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
for tag in ("ban", "circ"):
    m, zfs, xfs = seeded_head256(tag)
    m = m.to(dev)
    for cl in (False, True):
        z = [t.to(dev) for t in zfs]; x = [t.to(dev) for t in xfs]
        if cl: x = [t.contiguous(memory_format=torch.channels_last) for t in x]
        for _ in range(5): m(z, x)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10): m(z, x)
            torch.cuda.synchronize()
        agg = {}
        for e in prof.events():
            if e.device_type == torch.autograd.DeviceType.CUDA:
                a = agg.setdefault(e.name[:90], [0.0, 0]); a[0] += e.device_time; a[1] += 1
        tot = sum(t for t, c in agg.values())
        print(tag, "channels_last" if cl else "nchw", "device us per forward %.1f" % (tot / 10))
        for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
            print("   %-90s %7.1f us x%.1f" % (k, t / 10, c / 10))
