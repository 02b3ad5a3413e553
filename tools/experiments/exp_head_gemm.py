"""The packed head's first 1x1 convolution as a batched matrix product: hid = relu(b1 + W1 [6,256,256] @ feats [6,256,HW]) for HW = 169 (MultiCircBAN),
625 (MultiBAN) and 961 (configs[4]): which formulation does hipBLASLt run fastest at these sizes?  (10 products per hipGraph, us per product.)"""
import torch
dev = torch.device("cuda:0")
def graph_us(fn, inner=10, reps=20):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = [fn() for _ in range(inner)]
    for _ in range(3): g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * inner) * 1e3
for HW in (169, 625, 961):
    w = torch.randn(6, 256, 256, device=dev); f = torch.randn(6, 256, HW, device=dev); b = torch.randn(6, 256, 1, device=dev)
    wt = w.transpose(1, 2).contiguous(); bt = b.transpose(1, 2).contiguous()
    ref = torch.baddbmm(b, w, f)
    forms = {
        "baddbmm(b, W, f)": lambda: torch.baddbmm(b, w, f),
        "bmm(W, f) + b": lambda: torch.bmm(w, f).add_(b),
        "baddbmm(b^T, f^T, W^T) (transposed problem)": lambda: torch.baddbmm(bt, f.transpose(1, 2), wt),
        "bmm(f^T, W^T)": lambda: torch.bmm(f.transpose(1, 2), wt),
        "one matmul per head (6 launches)": lambda: [torch.addmm(b[i], w[i], f[i]) for i in range(6)],
        "W as one [1536,256] x per-head columns: matmul(Wcat, fcat) (6x the work)": None,
    }
    for name, fn in forms.items():
        if fn is None: continue
        out = fn()
        if isinstance(out, list): out = torch.stack(out)
        if out.shape != ref.shape: out = out.transpose(1, 2)
        err = float((out - ref).abs().max())
        print("HW=%4d  %-50s %7.1f us   max diff %.1e" % (HW, name, graph_us(fn), err))
