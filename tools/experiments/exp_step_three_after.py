"""After the 31x31 launch: the 13x13 and 5x5 launches back to back on one stream (shipping) vs on two streams, the head on a third."""
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
hs, cs = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

def head():
    feats = SF.share_feature(imgs2, folded).reshape(P, 2, 127, 127)
    Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
    pf = SF.share_feature(warped, folded)
    G.l1_score2(feats[0, 1], pf[0, 0], feats[0, 0], 1.0 / (127 * 127))

def step2():
    main = torch.cuda.current_stream()
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    hs.wait_stream(main)
    with torch.cuda.stream(hs): head()
    X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
    main.wait_stream(hs)

def step3(circ_first_on_side=True):
    main = torch.cuda.current_stream()
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    hs.wait_stream(main); cs.wait_stream(main)
    with torch.cuda.stream(hs): head()
    with torch.cuda.stream(cs): X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
    main.wait_stream(hs); main.wait_stream(cs)

def timed(fn, n=200):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(3):
    print(f"two streams (shipping) {timed(step2):.4f} ms   13x13 on its own stream {timed(step3):.4f} ms")
