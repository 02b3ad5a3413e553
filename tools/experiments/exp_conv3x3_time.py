"""B = 64 timing of the four stride-1 shapes of hdn_conv3x3_bias_relu_f32 under one hipGraph (20 launches per replay: no host
launch overhead in the number); used with tools/build_variant.sh ablation builds (HDN_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd.trunk import pack_conv3x3, conv3x3_bias_relu
dev = torch.device("cuda:0"); cl = torch.channels_last
B = int(os.environ.get("CV_B", "64"))
out = []
for (C, S) in ((64, 32), (128, 16), (256, 8), (512, 4)):
    w = torch.randn(C, C, 3, 3) * 0.05; b = torch.randn(C)
    wp = pack_conv3x3(w).to(dev); bd = b.to(dev)
    x = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl); r = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5): conv3x3_bias_relu(x, wp, bd, r)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): y = conv3x3_bias_relu(x, wp, bd, r)
    for _ in range(10): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    out.append("C=%d %.1f" % (C, e0.elapsed_time(e1) / 200 * 1e3))
print("%-28s B=%d  " % (os.path.basename(os.environ.get("HDN_LIB_PATH", "default")), B) + " | ".join(out))
