"""Full head at B = 64 as ONE batch on one stream vs n sub-batches on n streams (each layer's launch then has 256 / n workgroups and
the sub-batches' chains overlap each other's prologues / epilogues / launch gaps).  bench.build_full_head configuration."""
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
def timed(fn, n=50):
    for _ in range(8): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("1 stream, B = 64       %.3f ms/step" % timed(lambda: homo_stages(net, data)))
for ns in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    parts = [{k: v[i * 64 // ns:(i + 1) * 64 // ns].contiguous() for k, v in data.items()} for i in range(ns)]
    def run():
        main = torch.cuda.current_stream()
        outs = []
        for s, p in zip(streams, parts):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(homo_stages(net, p)["x"])
        for s in streams:
            main.wait_stream(s)
        return outs
    ref = homo_stages(net, data)["x"]
    got = torch.cat(run())
    torch.cuda.synchronize()
    print("%d streams, B = %d each  %.3f ms/step   max |x - x_single| = %.2e" % (ns, 64 // ns, timed(run), float((got - ref).abs().max())))
    # the same as one hipGraph (no host launch cost)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): run()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        keep = run()
    print("   as one hipGraph      %.3f ms/step" % timed(gr.replay))
