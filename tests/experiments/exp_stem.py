"""Fused trunk stem (conv7x7/s2 + folded BN + ReLU + maxpool3x3/s2): parity vs torch on the CPU, time vs the MIOpen stage."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd.trunk import resnet34_homo, fold_for_inference, FusedStem
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
torch.manual_seed(0)
base = resnet34_homo().eval()
for m in base.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
cpu = fold_for_inference(base, channels_last=False)
for (B, H, W) in ((2, 127, 127), (1, 9, 13), (3, 64, 128), (1, 1, 1), (2, 5, 7), (1, 126, 125)):
    x = torch.randn(B, 2, H, W)
    with torch.no_grad():
        ref = cpu.maxpool(cpu.relu(cpu.conv1(x)))
    for nhwc in (False, True):
        st = FusedStem(cpu.conv1, nhwc).to(dev)
        y = st(x.to(dev))
        assert y.shape == ref.shape, (y.shape, ref.shape)
        err = (y.cpu() - ref).abs().max().item()
        print((B, H, W), "nhwc" if nhwc else "nchw", "max err %.3g (|ref| max %.3g)" % (err, ref.abs().max().item()))
net = fold_for_inference(base.to(dev), channels_last=True)
fused = fold_for_inference(base.to(dev), channels_last=True, fused_stem=True)
x = torch.randn(64, 2, 127, 127, device=dev)
xl = x.contiguous(memory_format=torch.channels_last)
def t(f, n=50):
    with torch.no_grad():
        for _ in range(20): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("stem: MIOpen %.3f ms, fused %.3f ms; whole trunk: %.3f -> %.3f ms" % (
    t(lambda: net.maxpool(net.relu(net.conv1(xl)))), t(lambda: fused.conv1(x)), t(lambda: net(x.contiguous(memory_format=torch.channels_last))), t(lambda: fused(x))))
with torch.no_grad():
    a, b = net(xl), fused(x)
print("trunk output: max |fused - MIOpen| %.3g (|out| max %.3g)" % ((a - b).abs().max().item(), a.abs().max().item()))
# the kernel by itself (the module call above is bound by its Python overhead when nothing else is queued)
from hdn_amd import _lib
st = fused.conv1
out = torch.empty((64, 64, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
lib = _lib.load()
def raw():
    lib.hdn_trunk_stem_f32(_lib.ptr(x), _lib.ptr(st.wT), _lib.ptr(st.b), _lib.ptr(out), 64, 127, 127, 1, _lib.stream_ptr(dev))
for _ in range(10): raw()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): raw()
e1.record(); torch.cuda.synchronize()
print("kernel alone: %.1f us" % (e0.elapsed_time(e1) * 1000 / 50))
