"""hdn_conv3x3_bias_relu_f32 (split-bf16 implicit GEMM on the matrix cores) against MIOpen's fp32 convolution + the fused
epilogue, for the four stride-1 shapes of the trunk: error against a float64 convolution, and time at B = 64 and B = 1."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn.functional as F
from hdn_amd.trunk import pack_conv3x3, conv3x3_bias_relu, bias_relu_
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cl = torch.channels_last


def timed(fn, iters=50):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (C, S) in ((64, 32), (128, 16), (256, 8), (512, 4)):
    g = torch.Generator().manual_seed(C)
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    wp = pack_conv3x3(w).to(dev)
    wd = w.to(dev).contiguous(memory_format=cl)
    bd = b.to(dev)
    for B in (3, 64, 1):
        x = torch.randn(B, C, S, S, generator=g).clamp_min_(0)
        r = torch.randn(B, C, S, S, generator=g)
        xd, rd = x.to(dev).contiguous(memory_format=cl), r.to(dev).contiguous(memory_format=cl)
        y = conv3x3_bias_relu(xd, wp, bd, rd)
        y0 = conv3x3_bias_relu(xd, wp, bd)
        if B == 3:
            truth = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double())
            truth0 = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
            ref = torch.relu(F.conv2d(x, w, b, padding=1) + r)
            ym = torch.relu(F.conv2d(xd, wd, bd, padding=1) + rd).cpu()
            e = lambda t, tr: float((t.double() - tr).abs().max())
            print(f"C={C} S={S}: max err vs f64: ours {e(y.cpu(), truth):.2e} (no residual {e(y0.cpu(), truth0):.2e}) | PyTorch-CPU fp32 {e(ref, truth):.2e} | MIOpen fp32 {e(ym, truth):.2e}; max|y| {float(truth.abs().max()):.2f}")
        else:
            t_ours = timed(lambda: conv3x3_bias_relu(xd, wp, bd, rd))
            def lib():
                yy = F.conv2d(xd, wd, None, 1, 1)
                bias_relu_(yy, bd, rd)
            t_lib = timed(lib)
            t_conv = timed(lambda: F.conv2d(xd, wd, None, 1, 1))
            fl = 2.0 * B * C * C * 9 * S * S
            print(f"   B={B:2d}: ours {t_ours:6.1f} us ({fl / t_ours / 1e6:6.1f} TFLOP/s fp32-equivalent) | MIOpen conv {t_conv:6.1f} us + epilogue = {t_lib:6.1f} us")

# ---- the stride-2 convolution + downsample branch of the first block of layer2..4
from hdn_amd.trunk import pack_conv3x3s2_ds, conv3x3s2_ds
for (CI, S) in ((64, 16), (128, 8), (256, 4)):
    CO = 2 * CI
    g = torch.Generator().manual_seed(CI)
    w = torch.randn(CO, CI, 3, 3, generator=g) * (2.0 / (9 * CI)) ** 0.5
    wdn = torch.randn(CO, CI, 1, 1, generator=g) * (1.0 / CI) ** 0.5
    b = torch.randn(CO, generator=g) * 0.1
    wp = pack_conv3x3s2_ds(w, wdn).to(dev)
    wd_, wdd, bd = w.to(dev).contiguous(memory_format=cl), wdn.to(dev).contiguous(memory_format=cl), b.to(dev)
    for B in (64, 1):
        xd = torch.randn(B, CI, 2 * S, 2 * S, generator=g).clamp_min_(0).to(dev).contiguous(memory_format=cl)
        t_ours = timed(lambda: conv3x3s2_ds(xd, wp, bd))
        def lib():
            yy = F.conv2d(xd, wd_, None, 2, 1)
            bias_relu_(yy, bd)
            return F.conv2d(xd, wdd, None, 2)
        t_lib = timed(lib)
        print(f"s2 CI={CI} S={S} B={B:2d}: ours {t_ours:6.1f} us | MIOpen 3x3/s2 + epilogue + 1x1/s2 = {t_lib:6.1f} us")
