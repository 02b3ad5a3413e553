"""Experiment: is the 31x31 (x) 61x61 kernel clock/power limited?  Time it on random, constant and zero data."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hdn_amd
dev = torch.device("cuda:0")
def timeit(x, k, iters=20):
    for _ in range(3): hdn_amd.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): hdn_amd.xcorr_depthwise(x, k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, C = 64, 256
shape_x, shape_k = (B, C, 61, 61), (B, C, 31, 31)
flops = 2 * B * C * 31**4
for name, mk in (("relu(randn)", lambda s: torch.randn(s, device=dev).clamp_min_(0)), ("randn", lambda s: torch.randn(s, device=dev)),
                 ("ones", lambda s: torch.ones(s, device=dev)), ("zeros", lambda s: torch.zeros(s, device=dev))):
    x, k = mk(shape_x), mk(shape_k)
    ms = timeit(x, k)
    print(f"{name:12s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s")
