"""What does one MultiBAN / MultiCircBAN forward cost at the tracker's B = 1 with the production 256 channels (three levels, 7x7 template
features, 31x31 / 13x13 search features)?  hipGraph replay (no host launch overhead) + rocprof-friendly eager loop."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import heads as HD
dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
def run(cls, zs, xs, name):
    head = cls([256] * 3, 2, weighted=True).eval().to(dev)
    z = [torch.randn(1, 256, zs, zs, device=dev) for _ in range(3)]
    x = [torch.randn(1, 256, xs, xs, device=dev) for _ in range(3)]
    with torch.no_grad():
        for _ in range(20): head(z, x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = head(z, x)
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): g.replay()
        e1.record(); torch.cuda.synchronize()
        tg = e0.elapsed_time(e1) * 10
        t0 = time.perf_counter()
        for _ in range(100): head(z, x)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) * 1e4
    print(f"{name}: {tg:.0f} us per forward as a hipGraph, {te:.0f} us eager")
run(HD.MultiBAN, 7, 31, "MultiBAN 256 ch (z 7x7, x 31x31)")
run(HD.MultiCircBAN, 15, 15, "MultiCircBAN 256 ch (z 15x15, x 15x15)")
