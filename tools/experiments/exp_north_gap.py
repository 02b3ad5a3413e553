"""The 31x31 (x) 61x61 kernel back to back inside one hipGraph (bench.py's `sustained_launch_ms`, ~110 us) against the same launches with an
idle gap of G us between them (torch.cuda._sleep, same graph): (graph time - 10 G) / 10 per launch.  Separates "the chip clocks down under a
100 % duty cycle" from "a launch starts while its predecessor's 63 MB of output is still draining"."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(64, 256, 61, 61, generator=g).clamp_min_(0).to(dev)
k = torch.randn(64, 256, 31, 31, generator=g).clamp_min_(0).to(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(200): X.xcorr_depthwise(x, k)
torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 2_000_000 / (e0.elapsed_time(e1) * 1e3)
def graph_us(gap_us, n=10, reps=20, north=True):
    def body():
        for _ in range(n):
            if north: X.xcorr_depthwise(x, k)
            if gap_us: torch.cuda._sleep(int(gap_us * cyc_per_us))
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): body()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): body()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n
print("gap us | per launch incl. gap | gap alone (sleep kernel + its boundary) | launch = difference | frac of 8 TB/s")
for gap in (0, 1, 3, 6, 12, 25, 50, 100, 200):
    tot = graph_us(gap)
    alone = graph_us(gap, north=False) if gap else 0.0
    d = tot - alone
    print("%6d | %8.1f | %8.1f | %8.1f | %.3f" % (gap, tot, alone, d, 5778432 * 64 / (d * 1e-6) / 8e12))
