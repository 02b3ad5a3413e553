"""Experiment: order of the independent kernels inside the bench step vs the in-step duration of the 31x31 (x) 61x61 kernel."""
import itertools, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hdn_amd
from hdn_amd import xcorr as X, share_feature as SF, homography as G
import bench
dev = torch.device("cuda:0")
d = bench.make_inputs(dev, 0)
sf = hdn_amd.PreShareFeature().eval().to(dev); folded = sf.folded(dev)
imgs2 = d["imgs"].reshape(128, 1, 127, 127); tmpl = d["imgs"][:, :1].contiguous()
ev = []
def north(rec):
    if rec:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    if rec: e1.record(); ev.append((e0, e1))
def prod(rec): X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
def circ(rec): X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
def head(rec):
    feats = SF.share_feature(imgs2, folded).reshape(64, 2, 127, 127)
    Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
    pf = SF.share_feature(warped, folded)
    G.l1_score(feats[0, 1], pf[0, 0], 1.0 / 16129); G.l1_score(feats[0, 1], feats[0, 0], 1.0 / 16129)
parts = {"N": north, "P": prod, "C": circ, "H": head}
for order in (sys.argv[1:] or ("NPCH", "PNCH", "NPCH", "HNPC", "NPCH", "PCNH", "NCPH", "NHPC", "NPCH")):
    fs = [parts[c] for c in order]
    for _ in range(5):
        for f in fs: f(False)
    torch.cuda.synchronize(); ev.clear(); t = time.perf_counter()
    for _ in range(30):
        for f in fs: f(True)
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / 30 * 1e3
    nm = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    print(f"{order}: step {el:.3f} ms   north in-step {nm*1e3:.1f} us")
