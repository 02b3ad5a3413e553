"""Experiment: PyTorch-ROCm trunk variants at B=64 (ResNet-34 @127x127, fp32), one fresh process per variant:
    python tools/experiments/exp_trunk.py <benchmark 0|1> <asis|folded|nhwc>"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from hdn_amd.trunk import resnet34_homo, fold_for_inference
if len(sys.argv) < 3:
    import subprocess
    for b in ("0", "1"):
        for v in ("asis", "folded", "nhwc"):
            print(subprocess.run([sys.executable, __file__, b, v], capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
torch.backends.cudnn.benchmark = sys.argv[1] == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = resnet34_homo().eval().to(dev)
x = torch.randn(64, 2, 127, 127, device=dev)
v = sys.argv[2]
if v == "folded": net = fold_for_inference(net, channels_last=False)
if v == "nhwc": net = fold_for_inference(net, channels_last=True); x = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(5): net(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): net(x)
    torch.cuda.synchronize()
print(f"benchmark={sys.argv[1]} {v:7s} {(time.perf_counter()-t)/20*1e3:7.3f} ms")
