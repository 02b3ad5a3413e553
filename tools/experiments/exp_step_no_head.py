"""What the step would cost if the homography head were free: the three correlation launches alone on the main stream, against the shipping step
(head on its own stream from the end of the 31x31 launch) and against the head alone.  Bounds every 'make the head cheaper / move it' lever."""
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
hs = torch.cuda.Stream(device=dev)

def head():
    feats = SF.share_feature(imgs2, folded).reshape(P, 2, 127, 127)
    Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
    pf = SF.share_feature(warped, folded)
    G.l1_score2(feats[0, 1], pf[0, 0], feats[0, 0], 1.0 / (127 * 127))

def corr():
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])

def shipping():
    main = torch.cuda.current_stream()
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    hs.wait_stream(main)
    with torch.cuda.stream(hs): head()
    X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
    main.wait_stream(hs)

def timed(fn, n=200):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(3):
    print(f"shipping step {timed(shipping):.4f} ms   correlations only (no head at all) {timed(corr):.4f} ms   head alone {timed(head):.4f} ms", flush=True)
