#!/usr/bin/env python3
"""Round-5 verdict item 7: the first PreShareFeature launch of the configs[1] step (template + search images, 128 planes) issued at the START
of the step on a stream restricted to a few CUs (hipExtStreamCreateWithCUMask), so that it runs under the 31x31 launch's tail; DLT + warp,
the third PreShareFeature and the scores after the 31x31 launch on the ordinary head stream.  Against the shipping schedule
(--head-stream after-north), alternating, on one box.
    python tools/experiments/exp_step_cumask.py [steps]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench as BN
import hdn_amd
from hdn_amd import homography as G, share_feature as SF, xcorr as X

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
d = BN.make_inputs(dev, 0)
torch.manual_seed(BN.SEED)
sf = hdn_amd.PreShareFeature().eval().to(dev)
folded = sf.folded(dev)
PAIRS = BN.PAIRS
imgs2 = d["imgs"].reshape(PAIRS * 2, 1, 127, 127)
tmpl = d["imgs"][:, :1].contiguous()
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(n_cus, spread):
    words = (ctypes.c_uint32 * 8)()
    bits = [i * (256 // n_cus) for i in range(n_cus)] if spread else list(range(n_cus))
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


head_stream = torch.cuda.Stream(device=dev)


def step(mode, early=None):
    main = torch.cuda.current_stream()
    feats = None
    if early is not None:            # the first PreShareFeature under the 31x31 launch, on its few CUs
        early.wait_stream(main)
        with torch.cuda.stream(early):
            feats = SF.share_feature(imgs2, folded).reshape(PAIRS, 2, 127, 127)
    X.xcorr_depthwise(d["north_x"], d["north_k"])
    head_stream.wait_stream(main)
    if early is not None:
        head_stream.wait_stream(early)
    with torch.cuda.stream(head_stream):
        if feats is None:
            feats = SF.share_feature(imgs2, folded).reshape(PAIRS, 2, 127, 127)
        Hm, warped = G.dlt_warp(d["h4p"], d["off"], tmpl)
        pf = SF.share_feature(warped, folded)
        G.l1_score2(feats[0, 1], pf[0, 0], feats[0, 0], 1.0 / (127 * 127))
    X.xcorr_depthwise_multi(d["circ_x"], d["circ_k"], circular=True)
    X.xcorr_depthwise_multi(d["prod_x"], d["prod_k"])
    main.wait_stream(head_stream)


def timed(mode, early):
    for _ in range(20):
        step(mode, early)
    torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step(mode, early)
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    return wall


variants = [("shipping (after-north)", None)]
for n, spread in ((8, True), (16, True), (32, True), (32, False), (64, True)):
    try:
        variants.append((f"first PreShareFeature early on {n} CUs ({'spread over the chip' if spread else 'mask bits 0..%d' % (n - 1)})", masked_stream(n, spread)))
    except Exception as e:
        print("no masked stream:", e)
variants.append(("first PreShareFeature early, unmasked second stream", torch.cuda.Stream(device=dev)))
for rep in range(3):
    for name, early in variants:
        print(f"[{rep}] {name:<80s} {timed(name, early):.4f} ms per step", flush=True)
