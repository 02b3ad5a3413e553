import sys, os; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, bench
from hdn_amd.homo_model import homo_stages
from hdn_amd import trunk as T, _lib
from hdn_amd.trunk import pack_conv3x3, pack_conv3x3_v2, conv3x3_bias_relu
dev = torch.device("cuda:0"); cl = torch.channels_last
_lib.load().hdn_set_check_range(1)
for (C, S) in ((64, 32), (128, 16), (256, 8), (512, 4)):
  for B in (24, 33, 64):
    g = torch.Generator().manual_seed(7 * C + B)
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B, C, S, S, generator=g).clamp_min_(0)
    wp, wp2, bd = pack_conv3x3(w).to(dev), pack_conv3x3_v2(w).to(dev), b.to(dev)
    xd = x.to(dev).contiguous(memory_format=cl)
    y = conv3x3_bias_relu(xd, wp, bd, wpacked_v2=wp2)
torch.cuda.synchronize()
print("v2 calls done")
g = torch.Generator().manual_seed(bench.SEED)
imgs = torch.randn(64, 2, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(64, 1).to(dev)
net, data, _ = bench.build_full_head(dev, imgs, h4p)
orig = T.conv3x3_bias_relu
def wrapped(x, wp, b, residual=None, wpacked_v2=None):
    print("conv in", tuple(x.shape), "max|x| %.3e" % float(x.abs().max()), "finite", bool(torch.isfinite(x).all()), "ptr %x" % x.data_ptr(), flush=True)
    return orig(x, wp, b, residual, wpacked_v2)
T.conv3x3_bias_relu = wrapped
try:
    homo_stages(net, data)
except Exception as e:
    print("EXC", e)
