"""PreShareFeature rows kernel: time vs batch size (default strip height) to look for pathological sizes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import hdn_amd
from hdn_amd import share_feature as SF
dev = torch.device("cuda:0")
sf = hdn_amd.PreShareFeature().eval().to(dev); folded = sf.folded(dev)
def t(B, reps=50):
    x = torch.randn(B, 1, 127, 127, device=dev)
    for _ in range(10): SF.share_feature(x, folded)
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): SF.share_feature(x, folded)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / reps)
    return best
Bs = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 48, 64, 96, 100, 127, 128, 129, 160, 192, 256, 384, 512]
for B in Bs:
    print("B=%d: %.1f us" % (B, t(B)), flush=True)
