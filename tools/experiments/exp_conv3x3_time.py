import os, sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from hdn_amd.trunk import pack_conv3x3, conv3x3_bias_relu
dev = torch.device("cuda:0"); cl = torch.channels_last
def timed(fn, iters=50):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = []
for (C, S) in ((64, 32), (128, 16), (256, 8), (512, 4)):
    w = torch.randn(C, C, 3, 3) * 0.05; b = torch.randn(C)
    wp = pack_conv3x3(w).to(dev); bd = b.to(dev)
    x = torch.randn(64, C, S, S).to(dev).contiguous(memory_format=cl); r = torch.randn(64, C, S, S).to(dev).contiguous(memory_format=cl)
    out.append("C=%d %.1f" % (C, timed(lambda: conv3x3_bias_relu(x, wp, bd, r))))
print(os.path.basename(os.environ.get("HDN_LIB_PATH", "default")), " | ".join(out))
