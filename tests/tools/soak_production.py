#!/usr/bin/env python3
"""Soak of the drop-in tracker in its DEFAULT configuration (hipGraph per frame, folded backbone, MIOpen find mode) with the production-shaped model:
three sequences back to back on one DeviceTrackerHomo (re-capture per init), N frames each: ms per frame, device-memory growth between sequences
(must be 0 after the first), finite corners.    python tests/tools/soak_production.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np, torch
import sequence_bench as SB
from synth_sequence import make_sequence, LONG_WALK
from hdn_amd.tracker import DeviceTrackerHomo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
frames, corners, init = make_sequence(n_frames=60, frame_hw=(720, 1280), target_wh=(300, 200), **LONG_WALK)
model, _ = SB.build_production_model(frames, init, dev)
trk = DeviceTrackerHomo(model)
assert trk.use_graph and trk.folded == ["backbone", "neck", "neck_lp"] and trk.miopen_find and not torch.backends.cudnn.benchmark
mem = []
for seq in range(3):
    trk.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
    for i in range(5):
        trk.track_new(i, frames[1 + i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        out = trk.track_new(i, frames[1 + i % 59])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    mem.append(torch.cuda.memory_allocated())
    print("sequence %d: %d frames, %.3f ms per frame, graph %s, finite %s, allocated %.1f MB" % (seq, n, dt / n * 1e3, trk._graph is not None, bool(np.isfinite(np.asarray(out["points"])).all()), mem[-1] / 1e6), flush=True)
print("memory growth after the first sequence:", mem[2] - mem[1], "B")
sys.exit(0 if mem[2] == mem[1] else 1)
