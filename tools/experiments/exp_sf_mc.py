"""PreShareFeature: the matrix-core form (share_feature_mc.hip, HDN_SF_MC=1) against the rows-in-registers kernel: parity vs the CPU oracle and
kernel-level timing (run under rocprofv3 for the kernel times; wall times printed here include the launch path)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import hdn_amd
from hdn_amd import share_feature as SF
dev = torch.device("cuda:0")
torch.manual_seed(3)
m = hdn_amd.PreShareFeature().eval()
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d):
        mod.running_mean.uniform_(-0.5, 0.5); mod.running_var.uniform_(0.5, 2.0); mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.uniform_(-0.3, 0.3)
folded = SF.fold_params(m.state_dict()).to(dev)
for B in (128, 64):
    x = torch.randn(B, 1, 127, 127)
    xd = x.to(dev)
    y = SF.share_feature(xd, folded)
    if os.environ.get("CHECK"):
        with torch.no_grad():
            ref = m.ShareFeature(x[:4])
        print("B=%d max |y - torch| over 4 images: %.3e (max |ref| %.2f)" % (B, float((y[:4].cpu() - ref).abs().max()), float(ref.abs().max())), flush=True)
    for _ in range(20): SF.share_feature(xd, folded)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): SF.share_feature(xd, folded)
    torch.cuda.synchronize()
    print("B=%d HDN_SF_MC=%s: %.1f us per call (wall)" % (B, os.environ.get("HDN_SF_MC", "default"), (time.perf_counter() - t0) * 1e4), flush=True)
