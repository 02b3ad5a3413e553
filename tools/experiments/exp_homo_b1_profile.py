"""Device time per kernel of the homography estimator at the tracker's B = 1 (homo_refine: PreShareFeature, stem, 36 trunk convolutions, tail, DLT + warp,
PreShareFeature, scores, refinement warp), torch.profiler over graph replays."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "tools"))
import torch
from torch.profiler import profile, ProfilerActivity
import sequence_bench as SB
from hdn_amd.refine import homo_refine
dev = torch.device("cuda:0")
net = SB.seeded_net(0.1).to(dev)
net.optimize_for_inference(channels_last=True)
t = torch.randn(1, 1, 127, 127, device=dev); s = torch.randn(1, 1, 127, 127, device=dev)
fn = lambda: homo_refine(net, t, s, iterations=1)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): fn()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    keep = fn()
for _ in range(5): g.replay()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): g.replay()
b.record(); torch.cuda.synchronize()
print("graph replay %.1f us per frame" % (a.elapsed_time(b) * 10))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
agg = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        k = e.name[:100]; x = agg.setdefault(k, [0.0, 0]); x[0] += e.device_time; x[1] += 1
tot = sum(v[0] for v in agg.values())
print("device time %.1f us per frame in %.0f launches" % (tot / 10, sum(v[1] for v in agg.values()) / 10))
for k, (tt, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
    print("  %-100s %7.1f us x%.1f" % (k, tt / 10, c / 10))
