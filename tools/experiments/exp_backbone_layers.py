#!/usr/bin/env python3
"""Which convolution of the production-shaped backbone is slow at batch n?  Per Conv2d of tests/production_standin.py's ResNet-50 (stride 8,
dilated), PyTorch-ROCm / MIOpen in find mode, channels-last fp32: mean ms over 5 forward passes (events around each module), FLOPs, TFLOP/s.
    python tools/experiments/exp_backbone_layers.py [n] [size] [nchw]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "tools")):
    sys.path.insert(0, p)
import torch
import sequence_bench as SB
from synth_sequence import make_sequence, LONG_WALK

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 255
nchw = len(sys.argv) > 3 and sys.argv[3] == "nchw"
dev = torch.device("cuda:0")
frames, corners, init = make_sequence(n_frames=4, frame_hw=(720, 1280), target_wh=(300, 200), **LONG_WALK)
torch.backends.cudnn.benchmark = True
model, _ = SB.build_production_model(frames, init, dev, nchw=nchw)
# (BatchNorm left unfolded here: the folded form calls F.conv2d on copies of these weights — the convolutions themselves are the same)
x = torch.randn(n, 3, size, size, device=dev)
if not nchw:
    x = x.contiguous(memory_format=torch.channels_last)
recs = {}
def pre(m, inp):
    e = torch.cuda.Event(enable_timing=True); e.record(); m._e0 = e
def post(m, inp, out):
    e = torch.cuda.Event(enable_timing=True); e.record()
    recs.setdefault(m._name, []).append((m._e0, e, tuple(inp[0].shape), tuple(out.shape)))
for name, m in model.backbone.named_modules():
    if isinstance(m, torch.nn.Conv2d):
        m._name = name
        m.register_forward_pre_hook(pre); m.register_forward_hook(post)
with torch.no_grad():
    for _ in range(3):
        model.feature_extractor(x)
    recs.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        model.feature_extractor(x)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
rows = []
mods = dict(model.backbone.named_modules())
for name, lst in recs.items():
    ms = sum(a.elapsed_time(b) for a, b, _, _ in lst) / len(lst)
    m = mods[name]; i, o = lst[0][2], lst[0][3]
    fl = 2.0 * o[0] * o[1] * o[2] * o[3] * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
    rows.append((ms, name, i, o, m.kernel_size, m.stride, m.dilation, fl))
tot_ms, tot_fl = sum(r[0] for r in rows), sum(r[7] for r in rows)
print(f"n={n} size={size} {'NCHW' if nchw else 'channels-last'}: backbone forward {wall:.3f} ms wall (eager, hooks), convolutions {tot_ms:.3f} ms, {tot_fl / 1e9:.1f} GFLOP = {tot_fl / n / 1e9:.2f} per image, "
      f"{tot_fl / tot_ms / 1e9:.1f} TFLOP/s over the convolutions (fp32 matrix peak 157)")
for ms, name, i, o, k, s, d, fl in sorted(rows, reverse=True)[:14]:
    print(f"  {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TFLOP/s  {name:<28s} in {i} out {o} k{k} s{s} d{d}")
