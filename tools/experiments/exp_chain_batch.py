"""The folded trunk chained (hdn_conv3x3_chain_f32) against unchained, per batch size: where CHAIN_MAX_BATCH belongs."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, R)
import torch
import hdn_amd.trunk as T
dev = torch.device("cuda:0")
torch.manual_seed(0)
fast = T.fold_for_inference(T.resnet34_homo().to(dev).eval(), channels_last=True, fused_stem=True, fused_epilogue=True)
T.CHAIN_MAX_BATCH = 1 << 30
def graph_us(fn, reps=200):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    for _ in range(5): g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / reps
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
    x = torch.randn(B, 2, 127, 127, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        T.FusedBasicBlock.chain_disabled = False
        c = graph_us(lambda: fast(x))
        T.FusedBasicBlock.chain_disabled = True
        u = graph_us(lambda: fast(x))
    print("B %3d  chained %7.1f us  unchained %7.1f us" % (B, c, u), flush=True)
