"""Experiment: step time of the bench step as a function of how long the GPU has been busy (clock / power ramp)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hdn_amd
from hdn_amd import xcorr as X, share_feature as SF, homography as G
import bench
dev = torch.device("cuda:0")
d = bench.make_inputs(dev, 0)
sf = hdn_amd.PreShareFeature().eval().to(dev); folded = sf.folded(dev)
imgs2 = d["imgs"].reshape(128, 1, 127, 127); tmpl = d["imgs"][:, :1].contiguous()
def step():
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
    X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    feats = SF.share_feature(imgs2, folded).reshape(64, 2, 127, 127)
    Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
    pf = SF.share_feature(warped, folded)
    G.l1_score(feats[0, 1], pf[0, 0], 1.0 / 16129); G.l1_score(feats[0, 1], feats[0, 0], 1.0 / 16129)
step(); torch.cuda.synchronize()
time.sleep(2.0)  # let the GPU go idle
t0 = time.perf_counter(); acc = []
for w in range(30):
    t = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    acc.append(((time.perf_counter() - t0) * 1e3, (time.perf_counter() - t) / 10 * 1e3))
print(" ".join("%.0fms:%.3f" % a for a in acc))
