"""The configs[1] step (after-north schedule) with the correlation stream at a higher HIP stream priority than the head stream."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
import hdn_amd
from hdn_amd import homography as G, share_feature as SF, xcorr as X
dev = torch.device("cuda:0")
d = bench.make_inputs(dev, 0)
torch.manual_seed(bench.SEED)
sf = hdn_amd.PreShareFeature().eval().to(dev)
folded = sf.folded(dev)
P = bench.PAIRS
imgs2 = d["imgs"].reshape(P * 2, 1, 127, 127); tmpl = d["imgs"][:, :1].contiguous()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")

def head():
    feats = SF.share_feature(imgs2, folded).reshape(P, 2, 127, 127)
    Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
    pf = SF.share_feature(warped, folded)
    G.l1_score2(feats[0, 1], pf[0, 0], feats[0, 0], 1.0 / (127 * 127))

def make(main_prio, head_prio):
    main = torch.cuda.Stream(device=dev, priority=main_prio)
    hs = torch.cuda.Stream(device=dev, priority=head_prio)
    def step():
        with torch.cuda.stream(main):
            X.xcorr_depthwise(d["north_x"], d["north_k"])
            hs.wait_stream(main)
            with torch.cuda.stream(hs):
                head()
            X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
            X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
            main.wait_stream(hs)
    return step

def timed(fn, n=200):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(3):
    print("  ".join(f"main {a} head {b}: {timed(make(a, b)):.4f} ms" for a, b in ((0, 0), (-1, 0), (0, -1))))
