"""Phase clocks of the shipping 31x31 (x) 61x61 kernel (xcorr_north_fft4_kernel) from an instrumented build
(-DHDN_FFT_DEBUG_CLOCKS [-DNF4_EXP_SOLO=n], HDN_LIB_PATH): s_memtime marks of the first 16 workers, 6 interior iterations."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from hdn_amd import _lib, xcorr as X
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(64, 256, 61, 61, generator=g).clamp_min_(0).to(dev)
k = torch.randn(64, 256, 31, 31, generator=g).clamp_min_(0).to(dev)
lib = _lib.load()
lib.hdn_debug_read_phases.argtypes = [ctypes.c_void_p]
with X.north_variant("fft"):
    for _ in range(300): X.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); X.xcorr_depthwise(x, k); e1.record(); torch.cuda.synchronize()
    ph = np.zeros(16 * 16 * 16, np.uint64)
    assert lib.hdn_debug_read_phases(ph.ctypes.data) == 0
    ph = ph.reshape(16, 16, 16).astype(np.int64)
    names = ["0 first pass: AGPR reads, fetch next, twiddle, 64-pt FFT, split, row writes", "1 column reads + wait", "2 column FFT of the search pair",
             "3 kernel row pass (AGPR reads, 2 pruned halves, writes)", "4 kernel column pass + product", "5 inverse column FFT + writes",
             "6 inverse row pass: reads, re-pack, FFT, un-shift, half swap", "7 wait for next pair's loads + output stores"]
    w = [i for i in range(16) if ph[i, 1, 0] > 0]
    d = np.diff(ph[w][:, 1:7, :9], axis=2).reshape(-1, 8)
    print(f"{os.path.basename(os.environ.get('HDN_LIB_PATH', 'libhdn_hip.so'))} [{X.last_variant()}]: event {e0.elapsed_time(e1)*1e3:.1f} us; workers with marks {w}")
    for i, n in enumerate(names):
        print(f"   {n:82s} {d[:, i].mean():8.0f} {d[:, i].min():8.0f} {d[:, i].max():8.0f}")
    tot = (ph[w][:, 1:7, 8] - ph[w][:, 1:7, 0]).reshape(-1)
    print(f"   {'iteration (marks 0..8)':82s} {tot.mean():8.0f} {tot.min():8.0f} {tot.max():8.0f}")
