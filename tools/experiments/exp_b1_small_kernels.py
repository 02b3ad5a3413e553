"""B = 1 launches of the estimator's two small kernels (PreShareFeature, first trunk stage), for rocprofv3 kernel times."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd.trunk import FusedStem
from hdn_amd import share_feature as SF
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "1"))
torch.manual_seed(1)
st = FusedStem(torch.nn.Conv2d(2, 64, 7, 2, 3), True).to(dev)
x = torch.randn(B, 2, 127, 127, device=dev)
m = SF.PreShareFeature().eval().to(dev)
folded = SF.fold_params(m.state_dict()).to(dev)
y = torch.randn(B, 1, 127, 127, device=dev)
big = torch.empty(64 << 20, device=dev)
for i in range(60):
    big.zero_()                 # (256 MB through the caches between the launches: every launch starts cold, as in a frame)
    st(x)
    SF.share_feature(y, folded)
torch.cuda.synchronize()
