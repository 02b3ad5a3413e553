"""The 13x13 circular launch (memory-bound) BESIDE the 31x31 launch (issue-bound at half the HBM roofline) instead of behind it: does the step shrink?
  A (shipping)  north | then: head on its stream, 13x13 + 5x5 on the main one
  B             13x13 on its own stream from the start of the step, beside north; then head beside 5x5
  C             13x13 on its own stream started a little after north (behind a tiny dependent kernel chain) - same as B here, kept for the order of issue
"""
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

def step_a():
    main = torch.cuda.current_stream()
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    hs.wait_stream(main)
    with torch.cuda.stream(hs): head()
    X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
    main.wait_stream(hs)

def step_b(circ_first=True):
    main = torch.cuda.current_stream()
    cs.wait_stream(main)
    if circ_first:
        with torch.cuda.stream(cs): X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    if not circ_first:
        with torch.cuda.stream(cs): X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    hs.wait_stream(main)
    with torch.cuda.stream(hs): head()
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
    print(f"A shipping {timed(step_a):.4f} ms   B 13x13 beside north (issued first) {timed(step_b):.4f} ms   C (issued after north) {timed(lambda: step_b(False)):.4f} ms", flush=True)
