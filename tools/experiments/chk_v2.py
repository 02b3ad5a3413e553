import os, sys
sys.path.insert(0, "/root/repo")
import torch
from hdn_amd.trunk import pack_conv3x3, pack_conv3x3_v2, conv3x3_bias_relu
dev = torch.device("cuda:0"); cl = torch.channels_last
for B in (32, 64, 40):
  for (C, S) in ((64, 32), (128, 16), (256, 8), (512, 4)):
    w = torch.randn(C, C, 3, 3) * 0.05; b = torch.randn(C)
    wp = pack_conv3x3(w).to(dev); wp2 = pack_conv3x3_v2(w).to(dev); bd = b.to(dev)
    x = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl); r = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl)
    y1 = conv3x3_bias_relu(x, wp, bd, r)
    errs = []
    for rep in range(5):
        y2 = conv3x3_bias_relu(x, wp, bd, r, wpacked_v2=wp2)
        errs.append(float((y1 - y2).abs().max()))
    print(B, C, S, ["%.2e" % e for e in errs])
