"""Why the 31x31 (x) 61x61 kernel takes 92 us inside the bench step and 110 us back to back: its duration against its DUTY CYCLE.
Each launch is followed by an idle gap (torch.cuda._sleep) so that the kernel occupies 100 / 50 / 33 / 20 % of the time; the launch is
bracketed by events, the GPU's shader clock and socket power are sampled from hwmon meanwhile (50 ms period).
    python tools/experiments/exp_north_duty.py          -> table on stdout (profiles/round5_north.txt)"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import xcorr as X
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(64, 256, 61, 61, generator=g).clamp_min_(0).to(dev)
k = torch.randn(64, 256, 31, 31, generator=g).clamp_min_(0).to(dev)
hw = (glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") or [None])[0]
def read(name):
    try:
        return float(open(os.path.join(hw, name)).read())
    except Exception:
        return float("nan")
samples, stop = [], threading.Event()
def sampler():
    while not stop.is_set():
        samples.append((read("freq1_input") / 1e6, (read("power1_average") if os.path.exists(os.path.join(hw or "", "power1_average")) else read("power1_input")) / 1e6))
        time.sleep(0.05)
for _ in range(300): X.xcorr_depthwise(x, k)
torch.cuda.synchronize()
# calibrate _sleep: cycles per microsecond
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 2_000_000 / (e0.elapsed_time(e1) * 1e3)
print("hwmon %s; torch.cuda._sleep: %.0f cycles per us" % (hw, cyc_per_us))
print("%-28s %10s %10s %10s %10s %10s" % ("duty cycle (target)", "mean us", "min us", "frac", "sclk MHz", "power W"))
for duty in (1.0, 0.5, 0.33, 0.2, 0.1):
    gap_us = 0.0 if duty >= 1.0 else 100.0 * (1.0 / duty - 1.0)
    ev = []
    del samples[:]; stop.clear()
    th = threading.Thread(target=sampler, daemon=True)
    if hw: th.start()
    t_end = time.time() + 2.5
    while time.time() < t_end:
        for _ in range(50):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); X.xcorr_depthwise(x, k); b.record()
            if gap_us: torch.cuda._sleep(int(gap_us * cyc_per_us))
            ev.append((a, b))
        torch.cuda.synchronize()
    stop.set()
    if hw: th.join()
    t = torch.tensor([a.elapsed_time(b) * 1e3 for a, b in ev[len(ev) // 3:]])      # steady state: the last two thirds
    s = samples[len(samples) // 3:]
    f = sum(v[0] for v in s) / max(1, len(s)); p = sum(v[1] for v in s) / max(1, len(s))
    print("%-28s %10.1f %10.1f %10.3f %10.0f %10.0f" % ("%.0f %% (gap %.0f us)" % (100 * duty, gap_us), float(t.mean()), float(t.min()),
                                                       5778432 * 64 / (float(t.mean()) * 1e-6) / 8e12, f, p))
