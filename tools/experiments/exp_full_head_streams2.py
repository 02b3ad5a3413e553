"""Round 5: the full head (bench.build_full_head, B = 64) as n sub-batches on n streams inside ONE hipGraph, the later streams started with a delay.
Why: the matrix pipes are capped chip-wide (a pure-MFMA loop of hdn_conv3x3_v2_f32 takes 9.4 us on 256 workgroups and 4.8 us on 128: the same
~1.5 PFLOP/s either way), so a launch's prologue / epilogue / boundary is matrix-pipe time nobody uses — unless another chain is in its MFMA loop then.
Two chains started together stay in lockstep (no gain); started half a layer apart they alternate."""
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
ref = homo_stages(net, data)["x"]
def graphed(run):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): run()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        keep = run()
    return gr, keep
print("1 stream, B = 64, eager               %.3f ms/step" % timed(lambda: homo_stages(net, data)), flush=True)
for ns in (2, 3, 4):
    if 64 % ns: 
        sizes = [22, 21, 21]
    else:
        sizes = [64 // ns] * ns
    bounds = [sum(sizes[:i]) for i in range(ns + 1)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    parts = [{k: v[bounds[i]:bounds[i + 1]].contiguous() for k, v in data.items()} for i in range(ns)]
    for delay in (0, 8000, 16000, 32000):
        def run():
            main = torch.cuda.current_stream()
            outs = []
            for i, (s, p) in enumerate(zip(streams, parts)):
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    if delay and i: torch.cuda._sleep(delay * i)
                    outs.append(homo_stages(net, p)["x"])
            for s in streams:
                main.wait_stream(s)
            return outs
        gr, keep = graphed(run)
        gr.replay(); torch.cuda.synchronize()
        err = float((torch.cat(keep) - ref).abs().max())
        t_ms = timed(gr.replay)
        del gr
        print("%d streams (B = %s), start delay %5d cycles x i: %.3f ms/step   max |x - x_single| = %.2e" % (ns, sizes, delay, t_ms, err), flush=True)
