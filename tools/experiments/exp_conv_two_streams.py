"""Does a second, independent chain of half-batch convolution launches fill the matrix-pipe time the first chain leaves unused?
One hipGraph: (a) one stream, 20 launches at B = 64; (b) two streams, 20 launches at B = 32 each (with / without a start offset);
(c) one stream, 20 launches at B = 32.  Pre-allocated outputs (no allocator in the capture)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import _lib
from hdn_amd.trunk import pack_conv3x3_v2
dev = torch.device("cuda:0"); cl = torch.channels_last
lib = _lib.load()
def conv(x, wp2, bd, r, out, ws):
    B, C, S, _ = x.shape
    rc = lib.hdn_conv3x3_v2_f32(_lib.ptr(x), _lib.ptr(wp2), _lib.ptr(bd), _lib.ptr(r), _lib.ptr(out), _lib.ptr(ws) if ws is not None else None,
                                ws.numel() * 4 if ws is not None else 0, B, S, C, 0, _lib.stream_ptr(dev))
    _lib.check(rc, "conv")
def graph_time(fn, reps=10):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N = 20
for (C, S) in ((64, 32), (128, 16), (256, 8)):
    w = torch.randn(C, C, 3, 3) * 0.05
    wp2 = pack_conv3x3_v2(w).to(dev); bd = torch.randn(C).to(dev)
    def bufs(B):
        x = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl); r = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl)
        nws = lib.hdn_conv3x3_v2_workspace_bytes(B, S, C)
        return x, r, torch.empty_like(x), (torch.empty(nws // 4, device=dev) if nws else None)
    full, h0, h1 = bufs(64), bufs(32), bufs(32)
    def chain(b):
        for _ in range(N): conv(b[0], wp2, bd, b[1], b[2], b[3])
    t_full = graph_time(lambda: chain(full))
    t_half = graph_time(lambda: chain(h0))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res = []
    for delay in (0, 6000, 12000, 20000):
        def two():
            main = torch.cuda.current_stream()
            s1.wait_stream(main); s2.wait_stream(main)
            with torch.cuda.stream(s1): chain(h0)
            with torch.cuda.stream(s2):
                if delay: torch.cuda._sleep(delay)
                chain(h1)
            main.wait_stream(s1); main.wait_stream(s2)
        res.append("%d: %.1f" % (delay, graph_time(two) / N))
    print("C=%3d: per layer of 64 pairs: one chain B=64 %.1f us | one chain B=32 alone %.1f us (x2 = %.1f) | two chains B=32, start offset (cycles): %s us" % (
        C, t_full / N, t_half / N, 2 * t_half / N, "  ".join(res)), flush=True)
