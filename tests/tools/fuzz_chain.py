#!/usr/bin/env python3
"""Randomised parity of the chained trunk convolutions (hdn_conv3x3_chain_f32 / hdn_conv3x3_finish_f32) against float64 and against the
unchained launches: random batch 1..16, any of the four stride-1 shapes or a stride-2 / downsample block in front, activation / lazy inputs,
residual as activation or slices, data scales.    python tests/tools/fuzz_chain.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.nn.functional as F
from hdn_amd.trunk import pack_conv3x3, pack_conv3x3s2_ds, conv3x3_bias_relu, conv3x3s2_ds, chain_conv, LazyAct
dev = torch.device("cuda:0"); cl = torch.channels_last
cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = np.random.default_rng(seed); g = torch.Generator().manual_seed(seed)
bad = 0
SHAPES = [(64, 32), (128, 16), (256, 8), (512, 4)]
for it in range(cases):
    C, S = SHAPES[int(r.integers(0, 4))]; B = int(r.integers(1, 17)); sx = float(10 ** r.uniform(-1.5, 1.2))
    wsc = lambda co, ci, k: torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    down = C > 64 and bool(r.integers(0, 2))          # a stride-2 / downsample block in front (input C / 2 at 2S)
    if down:
        CI = C // 2
        x = torch.randn(B, CI, 2 * S, 2 * S, generator=g).relu_() * sx
        w1, wd, b1 = wsc(C, CI, 3), wsc(C, CI, 1), torch.randn(C, generator=g) * 0.1 * sx
        xd = x.to(dev).contiguous(memory_format=cl); p1 = pack_conv3x3s2_ds(w1, wd).to(dev); b1d = b1.to(dev)
        s1, sd, _ = chain_conv(xd, p1, 2)
        y_t = F.conv2d(x.double(), w1.double(), b1.double(), stride=2, padding=1).relu(); idt_t = F.conv2d(x.double(), wd.double(), None, stride=2)
        y_u, idt_u = conv3x3s2_ds(xd, p1, b1d)
        lazy, res_lazy, res_u = LazyAct(s1, b1d), sd, idt_u
    else:
        x = torch.randn(B, C, S, S, generator=g).relu_() * sx
        w1, b1 = wsc(C, C, 3), torch.randn(C, generator=g) * 0.1 * sx
        xd = x.to(dev).contiguous(memory_format=cl); p1 = pack_conv3x3(w1).to(dev); b1d = b1.to(dev)
        s1, _, _ = chain_conv(xd, p1)
        y_t = F.conv2d(x.double(), w1.double(), b1.double(), padding=1).relu(); idt_t = x.double()
        y_u = conv3x3_bias_relu(xd, p1, b1d)
        lazy, res_lazy, res_u = LazyAct(s1, b1d), xd, xd
    w2, b2 = wsc(C, C, 3), torch.randn(C, generator=g) * 0.1 * sx
    p2, b2d = pack_conv3x3(w2).to(dev), b2.to(dev)
    s2, _, _ = chain_conv(lazy, p2)
    out_l = LazyAct(s2, b2d, res_lazy)
    w3 = wsc(C, C, 3); p3 = pack_conv3x3(w3).to(dev)
    s3, _, xo = chain_conv(out_l, p3, 1, want_x=True)          # the block's output: finished while the next convolution stages it, and written out
    got, got3 = out_l.finish(), LazyAct(s3, b1d if not down else b2d).finish()
    t = (F.conv2d(y_t, w2.double(), b2.double(), padding=1) + idt_t).relu()
    u = conv3x3_bias_relu(y_u, p2, b2d, res_u)
    t3 = F.conv2d(t, w3.double(), (b1 if not down else b2).double(), padding=1).relu()
    e, e3 = float((got.cpu().double() - t).abs().max()), float((got3.cpu().double() - t3).abs().max())
    tol, tol3 = 2e-5 * float(t.abs().max()) + 1e-30, 3e-5 * float(t3.abs().max()) + 1e-30
    ok = e <= tol and e3 <= tol3 and torch.equal(xo, got) and float((got - u).abs().max()) <= 2e-5 * float(t.abs().max())
    if not ok:
        bad += 1
        print("MISMATCH case", it, dict(C=C, S=S, B=B, down=down, sx=sx), e, tol, e3, tol3, torch.equal(xo, got), float((got - u).abs().max()))
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
