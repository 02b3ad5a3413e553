"""Soak run of the device tracker loop (-> tests/tools, not collected by pytest): 3,000 frames eager and as one hipGraph per frame after a
200-frame warm-up: ms per frame, growth of the allocated device memory (must be 0), finite corners, host reads."""
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np, torch
from synth_sequence import make_sequence
from test_gpu_parity import _seeded_net
from hdn_amd.tracker import HomoTracker
dev = torch.device("cuda:0")
frames, corners, init = make_sequence(n_frames=40, frame_hw=(720, 1280), target_wh=(300, 200), seed=3)
net = _seeded_net().to(dev)
net.fc.bias.data.mul_(0.05); net.fc.weight.data.mul_(0.05)
net.optimize_for_inference(channels_last=True)
for graph in (False, True):
    trk = HomoTracker(net, graph=graph)
    trk.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
    dframes = [torch.from_numpy(f).to(dev) for f in frames]
    for i in range(200):      # one-time allocations (graph pool, library workspaces, caches) happen here
        out = trk.track_new(i, dframes[1 + i % 39], sync=(i % 100 == 0))
    torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    for i in range(3000):
        out = trk.track_new(i, dframes[1 + i % 39], sync=(i % 100 == 0))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pts = out["points"].cpu().numpy() if hasattr(out["points"], "cpu") else out["points"]
    print("graph" if graph else "eager", "3000 frames: %.3f ms/frame, memory growth %d B, finite %s, host syncs %d" % (dt / 3000 * 1e3, torch.cuda.memory_allocated() - m0, bool(np.isfinite(np.asarray(pts)).all()), trk.host_syncs))
