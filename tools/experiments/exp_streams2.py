"""Experiment (warm clocks): north alone, then the remaining kernels of the step on 1 / 2 / 3 streams."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hdn_amd
from hdn_amd import xcorr as X, share_feature as SF, homography as G
import bench
dev = torch.device("cuda:0")
d = bench.make_inputs(dev, 0)
sf = hdn_amd.PreShareFeature().eval().to(dev); folded = sf.folded(dev)
imgs2 = d["imgs"].reshape(128, 1, 127, 127); tmpl = d["imgs"][:, :1].contiguous()
def north(): X.xcorr_depthwise(d["north_x"], d["north_k"])
def prod(): X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
def circ(): X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
def head():
    feats = SF.share_feature(imgs2, folded).reshape(64, 2, 127, 127)
    Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
    pf = SF.share_feature(warped, folded)
    G.l1_score2(feats[0, 1], pf[0, 0], feats[0, 0], 1.0 / 16129)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()
def serial(): north(); prod(); circ(); head()
def two():   # north | then prod on main, circ+head on s1
    north()
    s1.wait_stream(main)
    with torch.cuda.stream(s1): circ(); head()
    prod()
    main.wait_stream(s1)
def three():
    north()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): circ()
    with torch.cuda.stream(s2): head()
    prod()
    main.wait_stream(s1); main.wait_stream(s2)
def allpar():
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): circ(); head()
    with torch.cuda.stream(s2): prod()
    north()
    main.wait_stream(s1); main.wait_stream(s2)
def head_par():  # the homography head on its own stream beside the three correlation launches
    s1.wait_stream(main)
    with torch.cuda.stream(s1): head()
    north(); circ(); prod()
    main.wait_stream(s1)
def head_par2():  # head beside circ + prod only (the 31x31 kernel fills every SIMD's registers by itself)
    north()
    s1.wait_stream(main)
    with torch.cuda.stream(s1): head()
    circ(); prod()
    main.wait_stream(s1)
def timeit(f, n=100):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2: f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for name, f in (("serial", serial), ("head | north, circ, prod", head_par), ("north, then head | circ, prod", head_par2), ("serial", serial), ("north, then prod | circ+head", two), ("north, then prod | circ | head", three), ("everything parallel", allpar), ("serial", serial)):
    print(f"{name:34s} {timeit(f):.3f} ms")
