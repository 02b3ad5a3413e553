"""Per-worker lifetime of the FFT north kernels from an instrumented build (-DHDN_FFT_DEBUG_CLOCKS, HDN_LIB_PATH):
shader clocks (s_memtime) against the 100 MHz real-time counter -> the clock the kernel really runs at, when the
workers start and end, and how uneven they are.   usage: exp_wave_clocks.py <variant>"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from hdn_amd import _lib, xcorr as X
dev = torch.device("cuda:0")
v = sys.argv[1]
g = torch.Generator().manual_seed(1)
x = torch.randn(64, 256, 61, 61, generator=g).clamp_min_(0).to(dev)
k = torch.randn(64, 256, 31, 31, generator=g).clamp_min_(0).to(dev)
lib = _lib.load()
lib.hdn_debug_read_clocks.argtypes = [ctypes.c_void_p]
with X.north_variant(v):
    for _ in range(300): X.xcorr_depthwise(x, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); X.xcorr_depthwise(x, k); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(8 * 4096, np.uint64)
    assert lib.hdn_debug_read_clocks(buf.ctypes.data) == 0
    b = buf.reshape(4096, 8).astype(np.int64)
    b = b[b[:, 3] > 0]
    sh = (b[:, 1] - b[:, 0]).astype(np.float64); rt = (b[:, 3] - b[:, 2]) / 100.0
    st = (b[:, 2] - b[:, 2].min()) / 100.0; en = (b[:, 3] - b[:, 2].min()) / 100.0
    print(f"{v} [{X.last_variant()}] WPG={os.environ.get('HDN_FFT_WPG','default')}: event {e0.elapsed_time(e1)*1e3:.1f} us, workers {len(b)}")
    print(f"   worker life us: min {rt.min():.1f} mean {rt.mean():.1f} max {rt.max():.1f};  shader clocks per worker mean {sh.mean():.0f} -> {np.mean(sh / rt):.0f} MHz")
    print(f"   start us after first: p50 {np.percentile(st,50):.1f} p99 {np.percentile(st,99):.1f} max {st.max():.1f};  end: min {en.min():.1f} p50 {np.percentile(en,50):.1f} max {en.max():.1f}")
    q = np.percentile(rt, [1, 10, 50, 90, 99])
    print("   life percentiles 1/10/50/90/99:", " ".join(f"{t:.1f}" for t in q))
    mhz = sh / rt
    print(f"   per-worker clock MHz: min {mhz.min():.0f} p50 {np.median(mhz):.0f} max {mhz.max():.0f};  shader clocks per worker: min {sh.min():.0f} max {sh.max():.0f}")
    xcc = b[:, 4] & 0xf
    hw = b[:, 5]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh_id = (hw >> 12) & 1; se = (hw >> 13) & 7
    for xid in sorted(set(xcc.tolist())):
        m = xcc == xid
        print(f"   XCC {xid}: workers {m.sum():4d}  life mean {rt[m].mean():6.1f} max {rt[m].max():6.1f}  clock {np.mean(mhz[m]):.0f} MHz  shader clk mean {sh[m].mean():.0f}")
    key = xcc * 1000000 + se * 10000 + sh_id * 1000 + cu * 10 + simd
    import collections
    cnt = collections.Counter(key.tolist())
    print("   waves per (xcc,se,sh,cu,simd) histogram:", sorted(collections.Counter(cnt.values()).items()), " distinct SIMDs:", len(cnt))
    cukey = xcc * 1000000 + se * 10000 + sh_id * 1000 + cu * 10
    ccnt = collections.Counter(cukey.tolist())
    print("   waves per CU histogram:", sorted(collections.Counter(ccnt.values()).items()), " distinct CUs:", len(ccnt))
    # life vs waves on the same SIMD / CU
    per = np.array([cnt[kk] for kk in key.tolist()]); perc = np.array([ccnt[kk] for kk in cukey.tolist()])
    for n in sorted(set(per.tolist())):
        print(f"   workers on a SIMD holding {n} worker(s): life mean {rt[per == n].mean():.1f} us ({(per == n).sum()})")
    for n in sorted(set(perc.tolist())):
        print(f"   workers on a CU holding {n} worker(s): life mean {rt[perc == n].mean():.1f} us ({(perc == n).sum()})")
    if v == "fft":
        lib.hdn_debug_read_phases.argtypes = [ctypes.c_void_p]
        ph = np.zeros(16 * 16 * 16, np.uint64)
        assert lib.hdn_debug_read_phases(ph.ctypes.data) == 0
        ph = ph.reshape(16, 16, 16).astype(np.int64)[:, :8, :14]      # worker, iteration, mark
        d = np.diff(ph, axis=2)                                         # 13 phases
        names = ["0 stash x->LDS + fetch next", "1 row reads issue+wait", "2 x twiddle + row FFT", "3 split + row writes", "4 col reads + wait",
                 "5 k stash/reads issue + column FFT", "6 k row pass (2 halves)", "7 k col pass + product (2 halves)", "8 inverse col FFT + writes",
                 "9 inv row reads + wait", "10 repack + inv row FFT + untwiddle + out writes", "11 out reads + wait", "12 wait vm + stores"]
        m = d[:, 1:7, :].reshape(-1, 13)  # skip first/last iterations
        print("   phase clocks (16 workers x 6 iterations): mean / min / max")
        for i, n in enumerate(names):
            print(f"     {n:52s} {m[:, i].mean():8.0f} {m[:, i].min():8.0f} {m[:, i].max():8.0f}")
        tot = (ph[:, 1:7, 13] - ph[:, 1:7, 0]).reshape(-1)
        print(f"     iteration total (marks 0..13)                        {tot.mean():8.0f} {tot.min():8.0f} {tot.max():8.0f}")
