"""Backbone convolution + folded shift + ReLU at B = 1: F.conv2d + hdn_bias_relu_f32 (two launches) against aten::miopen_convolution_relu
(MIOpen's fusion plan, when it has one) per shape of the ResNet-50 / stride-8 backbone, as hipGraph replays."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R)
import torch, torch.nn.functional as F
from hdn_amd.trunk import bias_relu_
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
def graph_us(fn, reps=50):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    for _ in range(3): g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / reps
shapes = [  # (cin, cout, k, stride, pad, dil, side)
    (64, 64, 1, 1, 0, 1, 63), (64, 64, 3, 1, 1, 1, 63), (64, 256, 1, 1, 0, 1, 63), (256, 128, 1, 1, 0, 1, 63), (128, 128, 3, 2, 0, 1, 63),
    (128, 512, 1, 1, 0, 1, 31), (512, 128, 1, 1, 0, 1, 31), (128, 128, 3, 1, 1, 1, 31), (512, 256, 1, 1, 0, 1, 31), (256, 256, 3, 1, 2, 2, 31),
    (256, 1024, 1, 1, 0, 1, 31), (1024, 256, 1, 1, 0, 1, 31), (1024, 512, 1, 1, 0, 1, 31), (512, 512, 3, 1, 4, 4, 31), (512, 2048, 1, 1, 0, 1, 31),
    (2048, 512, 1, 1, 0, 1, 31), (512, 512, 3, 1, 4, 4, 15), (256, 256, 3, 1, 2, 2, 15), (1024, 256, 1, 1, 0, 1, 15), (512, 2048, 1, 1, 0, 1, 15)]
for cl in (True, False):
    for (ci, co, k, s, p, d, side) in shapes:
        x = torch.randn(1, ci, side, side, device=dev); w = torch.randn(co, ci, k, k, device=dev) * 0.05; b = torch.randn(co, device=dev)
        if cl:
            x, w = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        two = lambda: bias_relu_(F.conv2d(x, w, None, s, p, d), b)
        t2 = graph_us(two)
        try:
            one = lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [s, s], [p, p], [d, d], 1)
            err = float((one() - two()).abs().max())
            t1 = graph_us(one)
        except Exception as e:
            t1, err = float("nan"), str(e)[:60]
        print("%s ci %4d co %4d k %d s %d d %d side %2d   conv + hdn pass %7.1f us   miopen_convolution_relu %7.1f us   diff %s" % ("NHWC" if cl else "NCHW", ci, co, k, s, d, side, t2, t1, err), flush=True)
