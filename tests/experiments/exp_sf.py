"""PreShareFeature: parity vs the oracle on a few shapes, time for 128 and 64 images of 127x127 (HDN_SF_LDS=1: the LDS kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import hdn_amd
from hdn_amd import share_feature as SF
from oracle import hdn_oracle as O
dev = torch.device("cuda:0")
torch.manual_seed(0)
sf = hdn_amd.PreShareFeature().eval()
for m in sf.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.8, 1.2)
sd = {"ShareFeature." + k: v.clone() for k, v in sf.ShareFeature.state_dict().items()}
sf = sf.to(dev); folded = sf.folded(dev)
for (B, H, W) in ((3, 127, 127), (2, 7, 9), (1, 8, 128), (2, 33, 100)):
    x = torch.randn(B, 1, H, W)
    y = SF.share_feature(x.to(dev), folded).cpu()
    ref = O.share_feature(x, sd)
    print((B, H, W), "max err %.3g" % (y - ref).abs().max())
for B in (128, 64):
    x = torch.randn(B, 1, 127, 127, device=dev)
    for _ in range(3): SF.share_feature(x, folded)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): SF.share_feature(x, folded)
    e1.record(); torch.cuda.synchronize()
    print("B=%d: %.1f us" % (B, e0.elapsed_time(e1) * 1000 / 50))
