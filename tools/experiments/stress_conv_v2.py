"""Race hunt: hdn_conv3x3_v2_f32 is deterministic, so N launches on the same inputs must give bit-identical outputs — with other work in flight
around them (a second stream hammering memory) and changing memory layouts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd.trunk import pack_conv3x3_v2, pack_conv3x3, conv3x3_bias_relu
dev = torch.device("cuda:0"); cl = torch.channels_last
N = int(os.environ.get("STRESS_N", "300"))
side = torch.cuda.Stream()
junk = torch.empty(32 << 20, device=dev)
for B in (64, 33, 24):
  for (C, S) in ((64, 32), (128, 16), (256, 8), (512, 4)):
    w = torch.randn(C, C, 3, 3) * 0.05; b = torch.randn(C)
    wp = pack_conv3x3(w).to(dev); wp2 = pack_conv3x3_v2(w).to(dev); bd = b.to(dev)
    x = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl); r = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl)
    ref = conv3x3_bias_relu(x, wp, bd, r, wpacked_v2=wp2).clone()
    v1 = conv3x3_bias_relu(x, wp, bd, r)
    bad = 0
    for i in range(N):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                junk.normal_()
        y = conv3x3_bias_relu(x, wp, bd, r, wpacked_v2=wp2)
        if not torch.equal(y, ref):
            bad += 1
            print("  MISMATCH at launch", i, "max diff", float((y - ref).abs().max()), "non-finite", int((~torch.isfinite(y)).sum()))
    torch.cuda.synchronize()
    print("B=%d C=%d: %d launches, %d mismatches, max |v2 - v1| %.2e" % (B, C, N, bad, float((ref - v1).abs().max())), flush=True)
