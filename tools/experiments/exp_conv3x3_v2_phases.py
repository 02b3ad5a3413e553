"""Phase timeline of hdn_conv3x3_v2_f32 inside the full head (measurement build: tools/build_variant.sh v2TIME conv3x3.hip -DHDN_ABLATION
-DCV2_EXP_TIME; HDN_LIB_PATH=...): s_memtime stamps per workgroup (100 MHz constant clock on gfx950: 10 ns ticks) of the LAST v2 launch
of each shape, taken by running the trunk up to the layer of interest."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from hdn_amd import _lib
from hdn_amd.trunk import pack_conv3x3_v2, pack_conv3x3, conv3x3_bias_relu
dev = torch.device("cuda:0"); cl = torch.channels_last
lib = _lib.load()
lib.hdn_cv2_debug_times.restype = ctypes.c_int
lib.hdn_cv2_debug_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
B = 64
names = ["prod start", "prod chunk0 stored", "prod last barrier in", "prod partials visible", "prod stores done", "cons loop start", "cons loop end", "cons dump done"]
for (C, S) in ((64, 32), (128, 16), (256, 8), (512, 4)):
    w = torch.randn(C, C, 3, 3) * 0.05; b = torch.randn(C)
    wp = pack_conv3x3(w).to(dev); wp2 = pack_conv3x3_v2(w).to(dev); bd = b.to(dev)
    x = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl); r = torch.randn(B, C, S, S).to(dev).contiguous(memory_format=cl)
    big = torch.empty(64 << 20, device=dev)
    for rep in range(3):
        big.normal_()                      # flush the caches with something else, as the application's other layers do
        y = conv3x3_bias_relu(x, wp, bd, r, wpacked_v2=wp2)
        torch.cuda.synchronize()
    lib.hdn_cv2_debug_times(None, 1); torch.cuda.synchronize()
    big.normal_()
    y = conv3x3_bias_relu(x, wp, bd, r, wpacked_v2=wp2)
    torch.cuda.synchronize()
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    lib.hdn_cv2_debug_times(buf.data_ptr(), 0)
    t = buf.cpu().view(4096, 8)
    t = t[(t[:, 0] > 0)]
    t0 = t[:, 0].min()
    rel = (t - t0).double() * 0.01      # us
    print("C=%d S=%d: %d workgroups; us after the first workgroup's start, median [min .. max]" % (C, S, t.shape[0]))
    for i in (0, 1, 5, 6, 7, 2, 3, 4):
        v = rel[:, i]
        print("   %-24s %6.2f  [%6.2f .. %6.2f]" % (names[i], float(v.median()), float(v.min()), float(v.max())))
