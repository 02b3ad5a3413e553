#!/usr/bin/env python3
"""Per-layer table of the homography trunk on the GPU: every distinct convolution of the BN-folded NHWC
ResNet-34 at the tracker's size (127 x 127 crops -> 32 x 32 after the fused first stage), timed alone with events under MIOpen
find mode, next to the fused epilogue that follows it (hdn_bias_relu_f32).  B = 64 (BASELINE configs[1..2]) and B = 1 (the
tracker's per-frame call).      python tools/trunk_layers.py > profiles/roundN_trunk_layers.txt
"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdn_amd.trunk import bias_relu_

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
# (name, count per forward, Cin, Cout, k, stride, H_in)
LAYERS = [("layer1 3x3 64->64 @32", 6, 64, 64, 3, 1, 32),
          ("layer2.0 3x3/s2 64->128 @32->16", 1, 64, 128, 3, 2, 32), ("layer2.0 ds 1x1/s2 64->128", 1, 64, 128, 1, 2, 32),
          ("layer2 3x3 128->128 @16", 7, 128, 128, 3, 1, 16),
          ("layer3.0 3x3/s2 128->256 @16->8", 1, 128, 256, 3, 2, 16), ("layer3.0 ds 1x1/s2 128->256", 1, 128, 256, 1, 2, 16),
          ("layer3 3x3 256->256 @8", 11, 256, 256, 3, 1, 8),
          ("layer4.0 3x3/s2 256->512 @8->4", 1, 256, 512, 3, 2, 8), ("layer4.0 ds 1x1/s2 256->512", 1, 256, 512, 1, 2, 8),
          ("layer4 3x3 512->512 @4", 5, 512, 512, 3, 1, 4)]


def timed(fn, iters=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


for B in (64, 1):
    print(f"## B = {B}   (fp32, NHWC, MIOpen find mode; us per call, back to back)")
    print(f"{'convolution':38s} {'n':>2s} {'conv us':>8s} {'TFLOP/s':>8s} {'epilogue us':>11s} {'GB/s':>6s}   total us (n x (conv + epi))")
    tot_c = tot_e = 0.0
    for name, n, ci, co, k, st, h in LAYERS:
        x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device=dev)
        pad = k // 2
        y = F.conv2d(x, w, None, st, pad)
        r = torch.randn_like(y)
        ho = y.shape[2]
        flops = 2.0 * B * co * ci * k * k * ho * ho
        tc = timed(lambda: F.conv2d(x, w, None, st, pad))
        is_ds = " ds " in name
        te = 0.0 if is_ds else timed(lambda: bias_relu_(y, b, r))
        nbytes = 3 * y.numel() * 4
        print(f"{name:38s} {n:2d} {tc:8.1f} {flops / tc / 1e6:8.1f} {te:11.1f} {(nbytes / te / 1e3 if te else 0):6.0f}   {n * (tc + te):8.1f}")
        tot_c += n * tc
        tot_e += n * te
    print(f"{'sum over the 36 convolutions / 32 epilogues':38s}    {tot_c:8.1f} {'':8s} {tot_e:11.1f}")
