"""Launch time of the trunk's first stage (FusedStem at B = 64, 127 px, channels-last) through whichever library HDN_LIB_PATH names."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd.trunk import FusedStem
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "64"))
torch.manual_seed(1)
st = FusedStem(torch.nn.Conv2d(2, 64, 7, 2, 3), True).to(dev)
x = torch.randn(B, 2, 127, 127, device=dev)
for off in ((True, False) if os.environ.get("BOTH") else (False,)):
    st.mfma_disabled = off
    for _ in range(10): st(x)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): st(x)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print("B=%d stem %s: %.2f us per launch" % (B, "vector pipe" if off else "matrix cores", best), flush=True)
