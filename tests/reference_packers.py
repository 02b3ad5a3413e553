"""TEST INFRASTRUCTURE: the weight-stream layouts of the matrix-core kernels as torch reshapes / permutes — the packers hdn_amd shipped until ABI 9,
kept as an independent statement of the layouts.  tests/test_host_logic.py holds the library's C packers (csrc/pack.hip, ABI 10) to them bit for bit."""
import ctypes

import torch

_MC_SIDE = {64: 32, 128: 16, 256: 8, 512: 4}

def pack_stem_mfma(weight):
    """[64, 2, 7, 7] fp32 weights (BatchNorm folded in) -> the fragment-ordered stream of hdn_trunk_stem_mfma_f32 (include/hdn_hip.h):
    [7 k steps][2 n tiles][2 pieces][k half g][n][8] fp16 bit patterns, element j = w[32 tile + n][ci][ky][kx = j] with
    ci * 7 + ky = 2 step + g, zero at j = 7."""
    import torch

    if tuple(weight.shape) != (64, 2, 7, 7):
        raise ValueError(f"pack_stem_mfma takes [64, 2, 7, 7] weights, got {tuple(weight.shape)}")
    w8 = torch.zeros(64, 14, 8, dtype=torch.float32)
    w8[:, :, :7] = weight.detach().to(torch.float32).cpu().reshape(64, 14, 7)                # [co][r = ci * 7 + ky][kx]
    t = _split_f16(w8).reshape(SPLIT_PIECES, 2, 32, 7, 2, 8)                                 # [pc, tile, n, step, g, j]
    return t.permute(3, 1, 0, 4, 2, 5).contiguous().view(torch.int16).reshape(-1)            # [step, tile, pc, g, n, j]



SPLIT_PIECES = 2


def _split_f16(w):
    """fp32 -> two fp16 pieces, w = p0 + 2^-11 p1 (round-to-nearest-even each; the residual w - p0 and its product with 2^11 are
    exact in fp32): the split the kernel applies to the activations (conv3x3.hip, split2x2)."""
    import torch

    if float(w.abs().max()) >= 65504.0:
        raise ValueError("conv3x3 matrix-core kernel: weights beyond the fp16 range")
    p0 = w.to(torch.float16)
    p1 = ((w - p0.float()) * 2048.0).to(torch.float16)
    return torch.stack([p0, p1])


def _pack(w4, S, CI, stride):
    """w4 [CO, CI, 3, T] fp32 (T taps per kernel row) -> [CO / BN][CI / (16 KS)][3][T][KS][2 pieces][2][BN][8] int16 bit patterns (fp16)."""
    import ctypes

    import torch

    from hdn_amd import _lib

    bn, ks = ctypes.c_int(0), ctypes.c_int(0)
    if _lib.load().hdn_conv3x3_pack_info(S, CI, stride, ctypes.byref(bn), ctypes.byref(ks)) != 0:
        raise ValueError(f"no matrix-core kernel for {CI} input channels at output side {S}, stride {stride}")
    BN, KS = bn.value, ks.value
    CO, T = w4.shape[0], w4.shape[3]
    pieces = _split_f16(w4.detach().to(torch.float32).cpu())            # [2, CO, CI, ky, t]
    t = pieces.permute(0, 1, 3, 4, 2).reshape(SPLIT_PIECES, CO // BN, BN, 3, T, CI // (16 * KS), KS, 2, 8)   # [piece, nb, n, ky, t, chunk, ks, g, 8]
    t = t.permute(1, 5, 3, 4, 6, 0, 7, 2, 8).contiguous()               # [nb, chunk, ky, t, ks, piece, g, n, 8]
    return t.view(torch.int16)


def pack_conv3x3(weight):
    """[C, C, 3, 3] fp32 weights of a stride-1 convolution -> the layout hdn_conv3x3_bias_relu_f32 streams (include/hdn_hip.h).
    The side S is implied by C in the trunk (64 -> 32, 128 -> 16, 256 -> 8, 512 -> 4)."""
    C = weight.shape[0]
    if tuple(weight.shape) != (C, C, 3, 3):
        raise ValueError(f"pack_conv3x3 takes [C, C, 3, 3] weights, got {tuple(weight.shape)}")
    return _pack(weight, _MC_SIDE.get(C, 0), C, 1)


def pack_conv3x3_v2(weight):
    """[C, C, 3, 3] fp32 weights -> the fragment-ordered stream of hdn_conv3x3_v2_f32 (include/hdn_hip.h):
    [C / (32 NT)][chunk][k slice][tap][k step of the slice][n tile][piece][k half][n][8] fp16 bit patterns."""
    import ctypes

    import torch

    from hdn_amd import _lib

    C = weight.shape[0]
    if tuple(weight.shape) != (C, C, 3, 3):
        raise ValueError(f"pack_conv3x3_v2 takes [C, C, 3, 3] weights, got {tuple(weight.shape)}")
    wk, ks, nt = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    if _lib.load().hdn_conv3x3_v2_pack_info(_MC_SIDE.get(C, 0), C, ctypes.byref(wk), ctypes.byref(ks), ctypes.byref(nt)) != 0:
        raise ValueError(f"no large-batch matrix-core kernel for {C} channels")
    WK, KS, NT = wk.value, ks.value, nt.value
    pieces = _split_f16(weight.detach().to(torch.float32).cpu()).reshape(SPLIT_PIECES, C, C, 9)        # [piece, co, ci, tap]
    # co = nb * 32 NT + nt * 32 + n;  ci = chunk * 16 KS + (j * WK + slice) * 16 + g * 8 + e
    t = pieces.reshape(SPLIT_PIECES, C // (32 * NT), NT, 32, C // (16 * KS), KS // WK, WK, 2, 8, 9)    # [pc, nb, nt, n, ch, j, wk, g, e, tap]
    t = t.permute(1, 4, 6, 9, 5, 2, 0, 7, 3, 8).contiguous()                                           # [nb, ch, wk, tap, j, nt, pc, g, n, e]
    return t.view(torch.int16).reshape(-1)




def pack_conv3x3s2_ds(weight, ds_weight):
    """[2C, C, 3, 3] weights of the stride-2 convolution + [2C, C, 1, 1] weights of the block's downsample branch -> the 4-tap layout of
    hdn_conv3x3s2_ds_f32: the 1x1 weights ride as a 4th tap of the middle kernel row."""
    import torch

    CO, CI = weight.shape[0], weight.shape[1]
    if tuple(weight.shape) != (2 * CI, CI, 3, 3) or tuple(ds_weight.shape) != (2 * CI, CI, 1, 1):
        raise ValueError(f"pack_conv3x3s2_ds takes [2C, C, 3, 3] and [2C, C, 1, 1] weights, got {tuple(weight.shape)}, {tuple(ds_weight.shape)}")
    w4 = torch.zeros(CO, CI, 3, 4, dtype=torch.float32)
    w4[:, :, :, :3] = weight.detach().float().cpu()
    w4[:, :, 1, 3] = ds_weight.detach().float().cpu()[:, :, 0, 0]
    return _pack(w4, _MC_SIDE.get(CO, 0), CI, 2)


def pack_conv3x3s2_ds_v2(weight, ds_weight):
    """[2C, C, 3, 3] + [2C, C, 1, 1] fp32 weights -> the fragment-ordered stream of hdn_conv3x3s2_v2_f32 (include/hdn_hip.h):
    [2C / 64][C / 32 chunks][2 k steps][10 steps: nine taps + the downsample branch][2 n tiles][piece][k half][n][8] fp16 bit patterns."""
    import torch

    CO, CI = weight.shape[0], weight.shape[1]
    if tuple(weight.shape) != (2 * CI, CI, 3, 3) or tuple(ds_weight.shape) != (2 * CI, CI, 1, 1) or CI % 32 or CO % 64:
        raise ValueError(f"pack_conv3x3s2_ds_v2 takes [2C, C, 3, 3] and [2C, C, 1, 1] weights, got {tuple(weight.shape)}, {tuple(ds_weight.shape)}")
    w10 = torch.zeros(CO, CI, 10, dtype=torch.float32)
    w10[:, :, :9] = weight.detach().float().cpu().reshape(CO, CI, 9)
    w10[:, :, 9] = ds_weight.detach().float().cpu()[:, :, 0, 0]
    # co = nb * 64 + nt * 32 + n;  ci = chunk * 32 + wk * 16 + g * 8 + e
    t = _split_f16(w10).reshape(SPLIT_PIECES, CO // 64, 2, 32, CI // 32, 2, 2, 8, 10)      # [pc, nb, nt, n, chunk, wk, g, e, step]
    return t.permute(1, 4, 5, 8, 2, 0, 6, 3, 7).contiguous().view(torch.int16).reshape(-1)  # [nb, chunk, wk, step, nt, pc, g, n, e]



def _pack_w1(w1):
    """[G, H, H] fp32 (row = output channel) -> the layout hdn_head_tail_f32 streams (include/hdn_hip.h): two fp16 pieces
    (v = p0 + 2^-11 p1), MFMA A-fragment order [G][H / 32][H / 16][piece][lane = 32 * k half + row][8] as int16 bit patterns."""
    G, H, _ = w1.shape
    w = w1.detach().to(torch.float32)
    p0 = w.to(torch.float16)
    p1 = ((w - p0.float()) * 2048.0).to(torch.float16)
    t = torch.stack([p0, p1]).view(2, G, H // 32, 32, H // 16, 2, 8)        # [piece, g, m tile, row, k step, k half, j]
    return t.permute(1, 2, 4, 0, 5, 3, 6).contiguous().view(torch.int16)   # [g, m tile, k step, piece, k half, row, j]


def _pack_conv_search(ws):
    """n folded conv_search weights [CO, 256, 3, 3] fp32 -> the layout hdn_head_conv3x3_f32 streams (include/hdn_hip.h): two fp16 pieces,
    [n][CO / 32][4 chunks][4 k slices][9 taps][piece][lane = 32 * k half + output channel][8] as int16 bit patterns."""
    w = torch.stack([t.detach().to(torch.float32) for t in ws])                # [n, CO, CI, 3, 3]
    n, CO, CI = w.shape[0], w.shape[1], w.shape[2]
    p0 = w.to(torch.float16)
    p1 = ((w - p0.float()) * 2048.0).to(torch.float16)
    t = torch.stack([p0, p1]).reshape(2, n, CO // 32, 32, CI // 64, 4, 2, 8, 9)    # [piece, n, cb, m, chunk, k slice, k half, j, tap]
    return t.permute(1, 2, 4, 5, 8, 0, 6, 3, 7).contiguous().view(torch.int16)   # [n, cb, chunk, k slice, tap, piece, k half, m, j]


