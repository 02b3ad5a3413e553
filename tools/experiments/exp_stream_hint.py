"""The two 5x5 correlation launches (6 problems each, B = 64, C = 256) with / without nontemporal loads and stores
(HDN_LIB_PATH selects the build: tools/build_variant.sh hint<n> xcorr.hip -DHDN_STREAM_HINT=<n>)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
mk = lambda *s: torch.randn(*s, generator=g).clamp_min_(0).to(dev)
sets = {"5x5 (x) 29x29": ([mk(64, 256, 29, 29) for _ in range(6)], [mk(64, 256, 5, 5) for _ in range(6)]),
        "5x5 (x) 35x35": ([mk(64, 256, 35, 35) for _ in range(6)], [mk(64, 256, 5, 5) for _ in range(6)])}
out = []
for name, (xs, ks) in sets.items():
    for _ in range(200): X.xcorr_depthwise_multi(xs, ks)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): y = X.xcorr_depthwise_multi(xs, ks)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 50)
    ref = torch.nn.functional.conv2d(xs[5][63, 255].double().cpu()[None, None], ks[5][63, 255].double().cpu()[None, None])[0, 0]
    out.append(f"{name}: {best:.1f} us (err {float((y[5][63, 255].cpu() - ref).abs().max()):.1e})")
print(os.path.basename(os.environ.get("HDN_LIB_PATH", "libhdn_hip.so (hint 3)")), " | ".join(out))
