"""A 1x1 convolution on a channels-last activation IS a GEMM [pixels, CI] x [CI, CO]: F.conv2d (MIOpen) + hdn_bias_relu_f32 against
torch._addmm_activation (hipBLASLt, bias + ReLU in the GEMM's epilogue) and torch.mm + hdn_bias_relu_f32, per 1x1 shape of the
ResNet-50 / stride-8 backbone at B = 1, as hipGraph replays."""
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
shapes = [(64, 64, 63), (64, 256, 63), (256, 64, 63), (256, 128, 63), (128, 512, 31), (512, 128, 31), (512, 256, 31), (256, 1024, 31), (1024, 256, 31),
          (1024, 512, 31), (512, 2048, 31), (2048, 512, 31), (2048, 256, 31), (256, 1024, 15), (1024, 256, 15), (512, 2048, 15), (2048, 512, 15), (2048, 256, 15)]
for (ci, co, side) in shapes:
    x = (torch.randn(1, ci, side, side, device=dev)).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 1, 1, device=dev) * 0.05; b = torch.randn(co, device=dev)
    wcl = w.contiguous(memory_format=torch.channels_last)
    wt = w.view(co, ci).t().contiguous()          # [CI, CO]
    wn = w.view(co, ci).contiguous()              # [CO, CI] (mm with a transposed view)
    a2 = x.permute(0, 2, 3, 1).reshape(side * side, ci)   # a view of the channels-last storage
    assert a2.data_ptr() == x.data_ptr()
    conv = lambda: bias_relu_(F.conv2d(x, wcl), b)
    lt = lambda: torch._addmm_activation(b, a2, wt, use_gelu=False)
    lt2 = lambda: torch._addmm_activation(b, a2, wn.t(), use_gelu=False)
    mm = lambda: bias_relu_(torch.mm(a2, wt).view(1, side, side, co).permute(0, 3, 1, 2), b)
    ref = conv().permute(0, 2, 3, 1).reshape(side * side, co)
    out = []
    for name, fn in (("conv+pass", conv), ("addmm_act", lt), ("addmm_act(w^T view)", lt2), ("mm+pass", mm)):
        try:
            y = fn()
            y2 = y if y.dim() == 2 else y.permute(0, 2, 3, 1).reshape(side * side, co)
            err = float((y2 - ref).abs().max())
            out.append("%s %6.1f us (diff %.1e)" % (name, graph_us(fn), err))
        except Exception as e:
            out.append("%s failed: %s" % (name, str(e)[:50]))
    print("ci %4d co %4d side %2d   " % (ci, co, side) + "   ".join(out), flush=True)
