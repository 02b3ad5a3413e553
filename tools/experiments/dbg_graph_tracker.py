import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from synth_sequence import make_sequence, success_4pts_error
from test_gpu_parity import _seeded_net
from hdn_amd.tracker import HomoTracker
dev = torch.device("cuda:0")
frames, corners, init = make_sequence(n_frames=6, frame_hw=(360, 640), target_wh=(150, 100), seed=8)
net = _seeded_net().to(dev)
a, b = HomoTracker(net), HomoTracker(net, graph=True)
for t in (a, b):
    t.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
for i in range(1, len(frames)):
    pa, pb = a.track_new(i, frames[i])["points"], b.track_new(i, frames[i])["points"]
    print(i, success_4pts_error(pa, pb), float(a.last_score), float(b.last_score))
    print(a.H_total.cpu().numpy().round(6).tolist(), b.H_total.cpu().numpy().round(6).tolist())
