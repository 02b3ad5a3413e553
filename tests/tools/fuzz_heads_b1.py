#!/usr/bin/env python3
"""Randomised parity of the two B = 1 head kernels of round 4 against float64 (bounds as in tests/test_gpu_parity.py):
hdn_head_conv3x3_f32 (random Hi x Wi, level counts, channel-block counts, NCHW / channels-last, data scales) and hdn_head_tail_f32
(random pixel counts, hidden 128 / 256, level counts, output rows).    python tests/tools/fuzz_heads_b1.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from hdn_amd import heads as HD
dev = torch.device("cuda:0")
cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = np.random.default_rng(seed)
g = torch.Generator().manual_seed(seed)
bad = 0
for it in range(cases):
    if it % 2 == 0:
        Wi = int(r.integers(3, 60)); Hi = int(r.integers(3, 60)); n = int(r.integers(1, 5)); CO = 32 * int(r.integers(1, 9)); nhwc = bool(r.integers(0, 2))
        Wo = Wi - 2
        rows = min((63 + Wo - 1) // Wo + 3, Hi)
        if rows * Wi > 224:
            continue
        sx, sw = float(10 ** r.uniform(-2, 1.5)), float(10 ** r.uniform(-2.5, -0.5))
        xs = [torch.randn(1, 256, Hi, Wi, generator=g).relu_() * sx for _ in range(n)]
        ws = [torch.randn(CO, 256, 3, 3, generator=g) * sw for _ in range(n)]
        bs = [torch.randn(CO, generator=g) * sx * sw * 20 for _ in range(n)]
        pk = HD._PackedHead(); pk.wsp = HD._pack_conv_search([w.to(dev) for w in ws]); pk.bsp = torch.stack(bs).to(dev)
        xd = [x.to(dev).contiguous(memory_format=torch.channels_last) if nhwc else x.to(dev) for x in xs]
        got = HD.head_conv_search(xd, pk).cpu().double()
        for i in range(n):
            ref = torch.nn.functional.conv2d(xs[i].double(), ws[i].double(), bs[i].double()).relu()[0]
            ref32 = torch.nn.functional.conv2d(xs[i], ws[i], bs[i]).relu()[0].double()
            e_ref, scale, err = float((ref32 - ref).abs().max()), float(ref.abs().max()), float((got[i] - ref).abs().max())
            if not err <= 4 * e_ref + 1e-6 * scale:
                bad += 1; print("CONV FAIL", Hi, Wi, n, CO, nhwc, sx, sw, err, e_ref, scale, flush=True)
    else:
        H = int(r.choice([128, 256])); P = int(r.integers(1, 1300)); n = int(r.integers(1, 4 if H == 256 else 5)); oc, ol = int(r.integers(1, 9)), int(r.integers(1, 9))
        sx, sw = float(10 ** r.uniform(-2, 1.5)), float(10 ** r.uniform(-2.5, -0.5))
        feats = torch.randn(2 * n, H, P, 1, generator=g).relu_() * sx
        pk = HD._PackedHead()
        pk.w1 = (torch.randn(2 * n, H, H, generator=g) * sw).to(dev); pk.b1 = (torch.randn(2 * n, H, 1, generator=g) * sx * sw * 10).to(dev)
        om = max(oc, ol)
        if n * (H * 128 + 4 * H + 4 * om * H) + (H // 32) * 8 * 32 * 4 > 160 * 1024:
            continue
        pk.wf = (torch.randn(2, om, n * H, generator=g) * 0.05).to(dev); pk.bf = torch.randn(2, om, 1, generator=g).to(dev)
        pk.w1p = HD._pack_w1(pk.w1)
        got = HD.head_tail(feats.to(dev), pk, n).cpu().double()
        f64 = lambda t: t.detach().cpu().double()
        hid = torch.baddbmm(f64(pk.b1), f64(pk.w1), f64(feats).view(2 * n, H, -1)).relu()
        ref = torch.baddbmm(f64(pk.bf), f64(pk.wf), hid.view(2, n * H, -1))
        hid32 = torch.baddbmm(pk.b1.cpu(), pk.w1.cpu(), feats.view(2 * n, H, -1)).relu()
        ref32 = torch.baddbmm(pk.bf.cpu(), pk.wf.cpu(), hid32.view(2, n * H, -1)).double()
        e_ref, scale, err = float((ref32 - ref).abs().max()), float(ref.abs().max()), float((got - ref).abs().max())
        if not err <= 4 * e_ref + 1e-6 * scale:
            bad += 1; print("TAIL FAIL", H, P, n, oc, ol, sx, sw, err, e_ref, scale, flush=True)
print(f"{cases} cases, {bad} violations")
