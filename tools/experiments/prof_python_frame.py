import sys, cProfile, pstats, io
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np, torch
from synth_sequence import make_sequence
from test_gpu_parity import _seeded_net
from hdn_amd.tracker import HomoTracker
dev = torch.device("cuda:0")
frames, corners, init = make_sequence(n_frames=20, frame_hw=(720, 1280), target_wh=(300, 200), seed=3)
net = _seeded_net().to(dev)
net.fc.bias.data.mul_(0.05); net.fc.weight.data.mul_(0.05)
net.optimize_for_inference(channels_last=True)
trk = HomoTracker(net, graph=False)
trk.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
df = [torch.from_numpy(f).to(dev) for f in frames]
for i in range(100): trk.track_new(i, df[1 + i % 19], sync=False)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(300): trk.track_new(i, df[1 + i % 19], sync=False)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(22)
print(s.getvalue()[:4500])
