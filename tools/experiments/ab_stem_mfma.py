"""Same-box A/B of the trunk's first stage: hdn_trunk_stem_mfma_f32 (matrix cores, round 5) against hdn_trunk_stem_f32 (vector pipe),
alone (events around 50 launches) and inside the full head (bench.build_full_head, B = 64), alternating in one process."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from hdn_amd.homo_model import homo_stages
from hdn_amd.trunk import FusedStem
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(bench.SEED)
B = int(os.environ.get("B", "64"))
imgs = torch.randn(B, 2, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(B, 1).to(dev)
net, data, _ = bench.build_full_head(dev, imgs, h4p)
stems = [m for m in net._hdn_fast_trunk.modules() if isinstance(m, FusedStem)]
print("FusedStem modules:", len(stems), flush=True)
def set_off(off):
    for m in stems: m.mfma_disabled = off
x = torch.randn(B, 2, 127, 127, device=dev)
for off in (True, False, True, False):
    set_off(off)
    st = stems[0]
    for _ in range(5): st(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): st(x)
    e1.record(); torch.cuda.synchronize()
    print("stem alone, %s: %.2f us per launch (B = %d)" % ("vector pipe" if off else "matrix cores", e0.elapsed_time(e1) / 50 * 1e3, B), flush=True)
def timed(n=60):
    for _ in range(10): homo_stages(net, data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): homo_stages(net, data)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
xs = {}
for rep in range(4):
    for off in (True, False):
        set_off(off)
        ms = timed()
        xs.setdefault(off, []).append(ms)
        print("full head, stem on the %s: %.4f ms per %d pairs = %.1f k frames/s" % ("vector pipe" if off else "matrix cores", ms, B, B / ms), flush=True)
set_off(True); a = homo_stages(net, data)["x"]
set_off(False); b = homo_stages(net, data)["x"]
print("best: vector %.4f ms, matrix %.4f ms (%.1f %%); max |x_v - x_m| = %.2e (max |x| %.2e)" % (min(xs[True]), min(xs[False]), 100 * (min(xs[True]) / min(xs[False]) - 1), float((a - b).abs().max()), float(a.abs().max())))
