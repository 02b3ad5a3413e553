"""Is the in-step slowdown of the 31x31 (x) 61x61 kernel (110 us vs 99 us back to back) the Infinity Cache?  Back-to-back launches
on ONE input set vs rotating over 4 sets (1.2 GB: nothing survives in the 256 MB cache), and with a 600 MB memset in between."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import hdn_amd
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
sets = [(torch.relu(torch.randn(64, 256, 61, 61, device=dev)), torch.relu(torch.randn(64, 256, 31, 31, device=dev))) for _ in range(4)]
junk = torch.empty(150_000_000, device=dev)
def run(name, nsets, flush):
    for i in range(8): X.xcorr_depthwise(*sets[i % nsets])
    torch.cuda.synchronize()
    ev = []
    for i in range(40):
        if flush: junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); X.xcorr_depthwise(*sets[i % nsets]); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    print("%-40s median %.1f us  min %.1f" % (name, t[len(t) // 2], t[0]))
run("one input set, back to back", 1, False)
run("4 input sets in rotation", 4, False)
run("one set, 600 MB memset between launches", 1, True)
run("one input set, back to back", 1, False)
