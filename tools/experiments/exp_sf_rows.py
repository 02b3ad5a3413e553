import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, hdn_amd
from hdn_amd import share_feature as SF
dev = torch.device("cuda:0")
torch.manual_seed(0)
sf = hdn_amd.PreShareFeature().eval().to(dev); folded = sf.folded(dev)
for B in (128, 64):
    x = torch.randn(B, 1, 127, 127, device=dev)
    for _ in range(100): SF.share_feature(x, folded)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): SF.share_feature(x, folded)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 10)
    print("HDN_SF_ROWS", os.environ.get("HDN_SF_ROWS", "4"), "B", B, " ".join(f"{t:.1f}" for t in ts), "us", float(SF.share_feature(x, folded).sum()))
