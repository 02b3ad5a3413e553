"""Experiment: where does the full head's time go, and does the BN-folded NHWC trunk help inside it?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hdn_amd
from hdn_amd.homo_model import homo_stages, _regress
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = hdn_amd.HomoModelBuilder().eval().to(dev)
imgs = torch.randn(64, 2, 127, 127, device=dev)
data = {"org_imgs": imgs, "input_tensors": imgs, "h4p": torch.tensor([[0,0,0,127,127,127,127,0.]], device=dev).repeat(64,1),
        "patch_indices": torch.arange(127*127, dtype=torch.float32, device=dev).repeat(64,1)}
def timeit(f, n=20, w=10):
    with torch.no_grad():
        for _ in range(w): f()
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
feats = net.ShareFeature(imgs.reshape(128,1,127,127)).reshape(64,2,127,127)
print("features: min %.3f max %.3f zeros %.2f" % (float(feats.min()), float(feats.max()), float((feats==0).float().mean())))
rnd = torch.randn_like(feats)
net.optimize_for_inference(True)
fast = net._hdn_fast_trunk
for name, inp in (("ShareFeature output", feats), ("randn", rnd)):
    a = timeit(lambda: net.backbone(inp))
    cl = inp.contiguous(memory_format=torch.channels_last)
    b = timeit(lambda: fast(cl))
    c = timeit(lambda: fast(inp.contiguous(memory_format=torch.channels_last)))
    print(f"{name:22s}: backbone {a:.3f} ms | folded NHWC (pre-converted) {b:.3f} ms | folded NHWC (+convert) {c:.3f} ms")
net.optimize_for_inference(False); a = timeit(lambda: homo_stages(net, data))
net.optimize_for_inference(True); b = timeit(lambda: homo_stages(net, data))
print(f"homo_stages: as-is {a:.3f} ms, folded+NHWC {b:.3f} ms")
