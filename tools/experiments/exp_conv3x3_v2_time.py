"""B = 64 timing of the four stride-1 trunk shapes: hdn_conv3x3_bias_relu_f32 (round 4) against hdn_conv3x3_v2_f32 (round 5), 20 launches
per hipGraph replay (no host launch cost in the number), with and without the residual."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd.trunk import pack_conv3x3, pack_conv3x3_v2, conv3x3_bias_relu
dev = torch.device("cuda:0"); cl = torch.channels_last
B = int(os.environ.get("CV_B", "64"))
def timed(f):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5): f()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    for _ in range(10): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3
for (C, S) in ((64, 32), (128, 16), (256, 8), (512, 4)):
    w = torch.randn(C, C, 3, 3) * 0.05; b = torch.randn(C)
    wp = pack_conv3x3(w).to(dev); wp2 = pack_conv3x3_v2(w).to(dev); bd = b.to(dev)
    x = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl); r = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl)
    d = float((conv3x3_bias_relu(x, wp, bd, r) - conv3x3_bias_relu(x, wp, bd, r, wpacked_v2=wp2)).abs().max())
    print("C=%3d S=%2d B=%d  v1 %.1f / %.1f us   v2 %.1f / %.1f us (with / without residual)   max |v1 - v2| %.2e" % (
        C, S, B, timed(lambda: conv3x3_bias_relu(x, wp, bd, r)), timed(lambda: conv3x3_bias_relu(x, wp, bd)),
        timed(lambda: conv3x3_bias_relu(x, wp, bd, r, wpacked_v2=wp2)), timed(lambda: conv3x3_bias_relu(x, wp, bd, wpacked_v2=wp2)), d))
