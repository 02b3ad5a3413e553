"""Is the FFT north-star kernel power limited?  Same instruction stream on zeros / ones / random data (warm clocks)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
def t(x, k, n=50):
    for _ in range(300): X.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): X.xcorr_depthwise(x, k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for v in ("fft", "fft2w"):
    with X.north_variant(v):
        for name, gen in (("zeros", lambda s: torch.zeros(s, device=dev)), ("ones", lambda s: torch.ones(s, device=dev)),
                          ("relu(randn)", lambda s: torch.relu(torch.randn(s, device=dev))), ("randn", lambda s: torch.randn(s, device=dev))):
            x, k = gen((64, 256, 61, 61)), gen((64, 256, 31, 31))
            print(v, name, "%.1f us" % t(x, k))
