"""Experiment: how much of the folded NHWC ResNet-34 trunk (B=64, 2x127x127) is its first stage (conv1 7x7 s2 + ReLU + maxpool)?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hdn_amd.trunk import resnet34_homo, fold_for_inference
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = fold_for_inference(resnet34_homo().eval().to(dev), channels_last=True)
x = torch.randn(64, 2, 127, 127, device=dev).contiguous(memory_format=torch.channels_last)
def t(f, n=50):
    with torch.no_grad():
        for _ in range(20): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def stage1(): return net.maxpool(net.relu(net.conv1(x)))
y = stage1()
def rest():
    z = y
    for l in (net.layer1, net.layer2, net.layer3, net.layer4): z = l(z)
    return z
print("whole trunk %.3f ms; first stage %.3f ms; conv1 only %.3f ms; layers 1-4 %.3f ms" % (t(lambda: net(x)), t(stage1), t(lambda: net.conv1(x)), t(rest)))
