"""Experiment: overlap the FMA-bound 31x31 (x) 61x61 correlation with the HBM-bound kernels of the step on two streams."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hdn_amd
from hdn_amd import xcorr as X, share_feature as SF, homography as G
import bench
dev = torch.device("cuda:0")
d = bench.make_inputs(dev, 0)
sf = hdn_amd.PreShareFeature().eval().to(dev); folded = sf.folded(dev)
imgs2 = d["imgs"].reshape(128, 1, 127, 127); tmpl = d["imgs"][:, :1].contiguous()
def rest():
    X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
    X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    feats = SF.share_feature(imgs2, folded).reshape(64, 2, 127, 127)
    Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
    pf = SF.share_feature(warped, folded)
    G.l1_score(feats[0, 1], pf[0, 0], 1.0 / 16129); G.l1_score(feats[0, 1], feats[0, 0], 1.0 / 16129)
def north(): X.xcorr_depthwise(d["north_x"], d["north_k"])
side = torch.cuda.Stream()
def seq(): north(); rest()
def par():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side): north()
    rest()
    main.wait_stream(side)
def par2():  # HBM-bound kernels first in queue order
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    rest_first = True
    with torch.cuda.stream(side): rest()
    north()
    main.wait_stream(side)
def timeit(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print(f"sequential            {timeit(seq):.3f} ms")
print(f"north on side stream  {timeit(par):.3f} ms")
print(f"rest on side stream   {timeit(par2):.3f} ms")
print(f"north alone           {timeit(north):.3f} ms   rest alone {timeit(rest):.3f} ms")
