import sys; sys.path.insert(0, "/root/repo")
import torch, bench
from hdn_amd.homo_model import homo_stages
from hdn_amd import trunk as T, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(bench.SEED)
imgs = torch.randn(64, 2, 127, 127, generator=g).to(dev)
h4p = torch.tensor([[0, 0, 0, 127, 127, 127, 127, 0]], dtype=torch.float32).repeat(64, 1).to(dev)
net, data, _ = bench.build_full_head(dev, imgs, h4p)
orig = T.conv3x3_bias_relu
def wrapped(x, wp, b, residual=None, wpacked_v2=None):
    print("conv in", tuple(x.shape), "max|x| %.3e" % float(x.abs().max()), "finite", bool(torch.isfinite(x).all()), "res max %.3e" % (float(residual.abs().max()) if residual is not None else 0))
    return orig(x, wp, b, residual, wpacked_v2)
T.conv3x3_bias_relu = wrapped
_lib.load().hdn_set_check_range(1)
try:
    homo_stages(net, data)
except Exception as e:
    print("EXC", e)
