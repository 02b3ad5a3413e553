"""hdn_head_tail_f32 alone at the three pixel counts of the tracker (13x13, 25x25, 31x31), 10 launches per hipGraph replay; with HDN_LIB_PATH for ablation builds."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from hdn_amd import heads as HD
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
out = []
for P in (169, 625, 961):
    n, H, om = 3, 256, 2
    pk = HD._PackedHead()
    pk.w1 = torch.randn(2 * n, H, H, device=dev) * 0.05; pk.b1 = torch.randn(2 * n, H, 1, device=dev)
    pk.wf = torch.randn(2, om, n * H, device=dev) * 0.05; pk.bf = torch.randn(2, om, 1, device=dev)
    pk.w1p = HD._pack_w1(pk.w1)
    feats = torch.randn(2 * n, H, P, 1, device=dev)
    out.append("P=%d %.1f us" % (P, graph_us(lambda: HD.head_tail(feats, pk, n))))
print(os.path.basename(os.environ.get("HDN_LIB_PATH", "default")), " | ".join(out))
