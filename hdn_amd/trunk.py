"""The homography regressor trunk: a 2-channel-input ResNet-34 kept on PyTorch-ROCm (MIOpen).

Reference: homo_estimator/Deep_homography/Oneline_DLTv1/backbone/resnet.py:137-194 with
BasicBlock x [3,4,6,3], conv1 = Conv2d(2,64,7,2,3), used_layers=[4] (returns layer4 only).
Parameter names match the reference (conv1, bn1, layerN.M.{conv1,bn1,conv2,bn2,downsample.0,downsample.1})
so its state_dict loads unchanged.  This is a dense MFMA-bound contraction and legitimately a library job
(SURVEY.md §8f rank 4); it is plumbing here, not one of the hand-written kernels.
"""
from __future__ import annotations

import os

import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += idt
        return self.relu(y)


class HomoResNet(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), in_channels=2):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(64, layers[0], 1)
        self.layer2 = self._stage(128, layers[1], 2)
        self.layer3 = self._stage(256, layers[2], 2)
        self.layer4 = self._stage(512, layers[3], 2)
        for m in self.modules():  # resnet.py:154-159
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _stage(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        mods = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        mods += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    act_domain = 0       # 1 on a folded copy whose stages all run in the scaled domain (fold_for_inference): activations are x * 2^-ACT_SCALE_LOG2 inside

    def forward_scaled(self, x):
        """The trunk's output in ITS activation domain (x 2^-8 when act_domain == 1: what hdn_avgpool_fc_f32 takes with in_domain = 1)."""
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return x.finish() if isinstance(x, LazyAct) else x       # (the chained small-batch form of the folded trunk: FusedBasicBlock)

    def forward(self, x):
        x = self.forward_scaled(x)
        return x.mul_(float(1 << ACT_SCALE_LOG2)) if self.act_domain else x      # (a fresh tensor of this forward: in place is safe; exact)


# The matrix-core kernels split an activation as x * 2^-8 (csrc/mfma_split.h: finite and fp32-accurate to |x| < 1.67e7).  A fully fused trunk pays that
# multiply ONCE: its first stage writes relu(conv) * 2^-8, every block runs with act_domain = 1 (activations already scaled in memory, biases handed over
# scaled: exact), and the exit multiplies by 2^8 (hdn_avgpool_fc_f32's in_domain, or HomoResNet.forward).
ACT_SCALE_LOG2 = 8


def resnet34_homo():
    return HomoResNet((3, 4, 6, 3))


# hdn_trunk_stem_mfma_f32 from this batch on (measured, rocprofv3 kernel time, MI355X, matrix cores / vector pipe: B = 1 8.8 / 12.0 us with cold caches,
# 8: 10.6 / 12.8, 16: 10.8 / 18.9, 32: 12.5 / 31.9, 64: 17.4 / 53.0); HDN_STEM_MFMA_MIN_BATCH: A/B switch
STEM_MFMA_MIN_BATCH = int(os.environ.get("HDN_STEM_MFMA_MIN_BATCH", "1"))


def _c_pack(what, n_bytes, call):
    """Run one of the library's packers (csrc/pack.hip, host code): -> int16 CPU tensor holding the opaque stream."""
    import torch

    from . import _lib

    if n_bytes < 0:
        raise ValueError(f"{what}: no matrix-core kernel takes weights of this shape")
    out = torch.empty(n_bytes // 2, dtype=torch.int16)
    rc = call(out.data_ptr(), n_bytes)
    if rc == -3:                                   # HDN_E_LIMIT
        raise ValueError(f"{what}: weights beyond the fp16 range (|w| >= 65,504) or NaN")
    _lib.check(rc, what)
    return out


def _host_f32(t):
    import torch

    return t.detach().to(device="cpu", dtype=torch.float32).contiguous()


def pack_stem_mfma(weight):
    """[64, 2, 7, 7] fp32 weights (BatchNorm folded in) -> the stream hdn_trunk_stem_mfma_f32 takes (hdn_pack_stem_mfma_f32; the layout is
    the library's: csrc/pack.hip)."""
    from . import _lib

    if tuple(weight.shape) != (64, 2, 7, 7):
        raise ValueError(f"pack_stem_mfma takes [64, 2, 7, 7] weights, got {tuple(weight.shape)}")
    lib, w = _lib.load(), _host_f32(weight)
    return _c_pack("pack_stem_mfma", lib.hdn_pack_stem_mfma_bytes(), lambda o, n: lib.hdn_pack_stem_mfma_f32(w.data_ptr(), o, n))


class FusedStem(nn.Module):
    """conv1 + folded bn1 + relu + maxpool of the trunk as ONE HIP kernel (hdn_trunk_stem_f32): the 64-channel 64 x 64 conv output
    never goes through HBM.  Built from a folded conv (weight [64,2,7,7], bias [64]); eval / no-grad only; CUDA tensors only."""

    def __init__(self, conv: nn.Conv2d, channels_last: bool, out_domain: int = 0):
        super().__init__()
        self.out_domain = int(out_domain)        # 1: the output is relu(...) * 2^-ACT_SCALE_LOG2 (the scaled domain of a fully fused trunk)
        if tuple(conv.weight.shape) != (64, 2, 7, 7) or conv.stride != (2, 2) or conv.padding != (3, 3) or conv.bias is None:
            raise ValueError("FusedStem replaces Conv2d(2, 64, 7, 2, 3) with a folded bias")
        self.register_buffer("wT", conv.weight.detach().permute(1, 2, 3, 0).contiguous())  # [ci][ky][kx][co]
        self.register_buffer("b", conv.bias.detach().clone())
        self.register_buffer("wfrag", pack_stem_mfma(conv.weight).to(conv.weight.device))                           # the matrix-core form's weights (28 KB)
        self.channels_last = bool(channels_last)
        self.mfma_disabled = False                                                           # A/B switch (tools/experiments, tests)

    def forward(self, x):
        from . import _lib

        if x.dim() != 4 or x.shape[1] != 2 or x.dtype != self.wT.dtype:
            raise ValueError("FusedStem takes float32 [B,2,H,W]")
        B, _, H, W = x.shape
        if W < 2 or W > 128:  # outside the kernel's range: the same arithmetic through the library
            import torch.nn.functional as F

            y = F.conv2d(x, self.wT.permute(3, 0, 1, 2), self.b, stride=2, padding=3)
            y = F.max_pool2d(F.relu(y), 3, 2, 1)
            return y.mul_(2.0 ** -ACT_SCALE_LOG2) if self.out_domain else y
        dev = _lib.require_device(x, self.wT, self.b)
        if self.wfrag.device != dev:
            raise _lib.HdnHipError(f"FusedStem weights on {self.wfrag.device}, input on {dev}")
        xs = x.detach().contiguous()  # NCHW
        Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
        import torch

        out = torch.empty((B, 64, Hp, Wp), dtype=x.dtype, device=dev,
                          memory_format=torch.channels_last if self.channels_last else torch.contiguous_format)
        if self.channels_last and H == 127 and W == 127 and B >= STEM_MFMA_MIN_BATCH and not self.mfma_disabled:
            with _lib.device_guard(dev):
                rc = _lib.load().hdn_trunk_stem_mfma_f32(_lib.ptr(xs), _lib.ptr(self.wfrag), _lib.ptr(self.b), _lib.ptr(out), B, H, W, self.out_domain,
                                                         _lib.stream_ptr(dev))
            _lib.check(rc, "trunk_stem_mfma")
            return out
        with _lib.device_guard(dev):
            rc = _lib.load().hdn_trunk_stem_f32(_lib.ptr(xs), _lib.ptr(self.wT), _lib.ptr(self.b), _lib.ptr(out), B, H, W,
                                                1 if self.channels_last else 0, _lib.stream_ptr(dev))
        _lib.check(rc, "trunk_stem")
        return out.mul_(2.0 ** -ACT_SCALE_LOG2) if self.out_domain else out


def bias_relu_(y, bias, residual=None):
    """In place: y = relu(y + bias[c] (+ residual)) through hdn_bias_relu_f32; y / residual [B,C,H,W] float32, both NCHW-contiguous
    or both channels-last."""
    import torch

    from . import _lib

    dev = _lib.require_device(y, bias) if residual is None else _lib.require_device(y, bias, residual)
    if y.dim() != 4 or bias.numel() != y.shape[1] or (residual is not None and residual.shape != y.shape):
        raise ValueError(f"bias_relu_: y [B,C,H,W], bias [C], residual like y; got {tuple(y.shape)}, {tuple(bias.shape)}")
    B, C, H, W = y.shape
    if y.is_contiguous():   # (a [B,C,1,1] tensor is both: NCHW arithmetic is right for it)
        nhwc = 0
    elif y.is_contiguous(memory_format=torch.channels_last):
        nhwc = 1
    else:
        raise ValueError("bias_relu_: y must be NCHW-contiguous or channels-last")
    if residual is not None and not (residual.is_contiguous(memory_format=torch.channels_last) if nhwc else residual.is_contiguous()):
        residual = residual.contiguous(memory_format=torch.channels_last if nhwc else torch.contiguous_format)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_bias_relu_f32(_lib.ptr(y), _lib.ptr(bias), _lib.ptr(residual) if residual is not None else None, B, C, H * W,
                                           nhwc, _lib.stream_ptr(dev))
    _lib.check(rc, "bias_relu")
    return y


SPLIT_PIECES = 2


def pack_conv3x3(weight):
    """[C, C, 3, 3] fp32 weights of a stride-1 convolution -> the stream hdn_conv3x3_bias_relu_f32 / hdn_conv3x3_chain_f32 take
    (hdn_pack_conv3x3_f32).  The side S is implied by C in the trunk (64 -> 32, 128 -> 16, 256 -> 8, 512 -> 4)."""
    from . import _lib

    C = weight.shape[0]
    if tuple(weight.shape) != (C, C, 3, 3):
        raise ValueError(f"pack_conv3x3 takes [C, C, 3, 3] weights, got {tuple(weight.shape)}")
    lib, w = _lib.load(), _host_f32(weight)
    return _c_pack("pack_conv3x3", lib.hdn_pack_conv3x3_bytes(C), lambda o, n: lib.hdn_pack_conv3x3_f32(w.data_ptr(), C, o, n))


def pack_conv3x3_v2(weight):
    """[C, C, 3, 3] fp32 weights -> the stream of hdn_conv3x3_v2_f32 (hdn_pack_conv3x3_v2_f32)."""
    from . import _lib

    C = weight.shape[0]
    if tuple(weight.shape) != (C, C, 3, 3):
        raise ValueError(f"pack_conv3x3_v2 takes [C, C, 3, 3] weights, got {tuple(weight.shape)}")
    lib, w = _lib.load(), _host_f32(weight)
    return _c_pack("pack_conv3x3_v2", lib.hdn_pack_conv3x3_v2_bytes(C), lambda o, n: lib.hdn_pack_conv3x3_v2_f32(w.data_ptr(), C, o, n))


V2_MIN_BATCH = 24      # below: the chained / K-sliced form of conv3x3_kernel (CHAIN_MAX_BATCH = 16 pairs and the sizes between)


def _s2_weights(weight, ds_weight, what):
    CO, CI = weight.shape[0], weight.shape[1]
    if tuple(weight.shape) != (2 * CI, CI, 3, 3) or tuple(ds_weight.shape) != (2 * CI, CI, 1, 1):
        raise ValueError(f"{what} takes [2C, C, 3, 3] and [2C, C, 1, 1] weights, got {tuple(weight.shape)}, {tuple(ds_weight.shape)}")
    return CI, _host_f32(weight), _host_f32(ds_weight)


def pack_conv3x3s2_ds(weight, ds_weight):
    """[2C, C, 3, 3] weights of the stride-2 convolution + [2C, C, 1, 1] weights of the block's downsample branch -> the stream of
    hdn_conv3x3s2_ds_f32 (hdn_pack_conv3x3s2_ds_f32)."""
    from . import _lib

    CI, w, wd = _s2_weights(weight, ds_weight, "pack_conv3x3s2_ds")
    lib = _lib.load()
    return _c_pack("pack_conv3x3s2_ds", lib.hdn_pack_conv3x3s2_ds_bytes(CI), lambda o, n: lib.hdn_pack_conv3x3s2_ds_f32(w.data_ptr(), wd.data_ptr(), CI, o, n))


def pack_conv3x3s2_ds_v2(weight, ds_weight):
    """The same two weight tensors -> the stream of hdn_conv3x3s2_v2_f32 (hdn_pack_conv3x3s2_v2_f32)."""
    from . import _lib

    CI, w, wd = _s2_weights(weight, ds_weight, "pack_conv3x3s2_ds_v2")
    lib = _lib.load()
    return _c_pack("pack_conv3x3s2_ds_v2", lib.hdn_pack_conv3x3s2_v2_bytes(CI), lambda o, n: lib.hdn_pack_conv3x3s2_v2_f32(w.data_ptr(), wd.data_ptr(), CI, o, n))


def conv3x3_bias_relu(x, wpacked, bias, residual=None, wpacked_v2=None, act_domain=0):
    """relu(conv3x3(x) + bias (+ residual)) through hdn_conv3x3_bias_relu_f32 — or, given `wpacked_v2` (pack_conv3x3_v2) and a batch of
    V2_MIN_BATCH or more, through hdn_conv3x3_v2_f32; x / residual channels-last [B,C,S,S] float32.  act_domain = 1: x, residual and the result are
    x_real * 2^-8 in memory and `bias` is bias * 2^-8 (include/hdn_hip.h, "Activation domain")."""
    import torch

    from . import _lib

    dev = _lib.require_device(x, bias) if residual is None else _lib.require_device(x, bias, residual)
    B, C, S, S2 = x.shape
    cl = torch.channels_last
    if S != S2 or not x.is_contiguous(memory_format=cl) or (residual is not None and (residual.shape != x.shape or not residual.is_contiguous(memory_format=cl))):
        raise ValueError("conv3x3_bias_relu: square channels-last inputs of equal shape")
    if wpacked.dtype != torch.int16 or wpacked.device != dev or wpacked.numel() != 9 * SPLIT_PIECES * C * C or bias.numel() != C:
        raise ValueError("conv3x3_bias_relu: weights must come from pack_conv3x3 for this channel count, on the input's device")
    out = torch.empty_like(x, memory_format=cl)
    lib = _lib.load()
    v2 = wpacked_v2 is not None and B >= V2_MIN_BATCH
    if v2 and (wpacked_v2.dtype != torch.int16 or wpacked_v2.device != dev or wpacked_v2.numel() != 9 * SPLIT_PIECES * C * C):
        raise ValueError("conv3x3_bias_relu: wpacked_v2 must come from pack_conv3x3_v2 for this channel count, on the input's device")
    nws = lib.hdn_conv3x3_v2_workspace_bytes(B, S, C) if v2 else lib.hdn_conv3x3_workspace_bytes(B, S, C, 1)
    if nws < 0:
        _lib.check(int(nws), "conv3x3_bias_relu")
    ws = torch.empty(nws // 4, dtype=torch.float32, device=dev) if nws else None   # (from torch's caching allocator: no sync, graph-safe)
    with _lib.device_guard(dev):
        fn = lib.hdn_conv3x3_v2_f32 if v2 else lib.hdn_conv3x3_bias_relu_f32
        rc = fn(_lib.ptr(x), _lib.ptr(wpacked_v2 if v2 else wpacked), _lib.ptr(bias), _lib.ptr(residual) if residual is not None else None,
                _lib.ptr(out), _lib.ptr(ws) if ws is not None else None, nws, B, S, C, int(act_domain), _lib.stream_ptr(dev))
    _lib.check(rc, "conv3x3_bias_relu")
    return out


# channel counts whose stride-1 3x3 convolutions run on hdn_conv3x3_bias_relu_f32 instead of MIOpen (measured per shape at
# B = 64, profiles/round3_conv3x3.txt: the kernel is kept only where it wins)
MATRIX_CORE_CHANNELS = (64, 128, 256, 512)
_MC_SIDE = {64: 32, 128: 16, 256: 8, 512: 4}


def conv3x3s2_ds(x, wpacked, bias, wpacked_v2=None, act_domain=0):
    """(relu(conv3x3/s2(x) + bias), conv1x1/s2(x)) through hdn_conv3x3s2_ds_f32 - or, given `wpacked_v2` (pack_conv3x3s2_ds_v2) and a batch of
    V2_MIN_BATCH or more, through hdn_conv3x3s2_v2_f32; x channels-last [B,C,2S,2S] -> two [B,2C,S,S]."""
    import torch

    from . import _lib

    dev = _lib.require_device(x, bias)
    B, CI, H, W = x.shape
    cl = torch.channels_last
    if H != W or H % 2 or not x.is_contiguous(memory_format=cl):
        raise ValueError("conv3x3s2_ds: square, even-sided channels-last input")
    S, CO = H // 2, 2 * CI
    if wpacked_v2 is not None and B >= V2_MIN_BATCH:
        if wpacked_v2.dtype != torch.int16 or wpacked_v2.device != dev or wpacked_v2.numel() != SPLIT_PIECES * 10 * CI * CO or bias.numel() != CO:
            raise ValueError("conv3x3s2_ds: v2 weights must come from pack_conv3x3s2_ds_v2 for this channel count, on the input's device")
        out = torch.empty((B, CO, S, S), dtype=torch.float32, device=dev, memory_format=cl)
        out_ds = torch.empty_like(out, memory_format=cl)
        with _lib.device_guard(dev):
            rc = _lib.load().hdn_conv3x3s2_v2_f32(_lib.ptr(x), _lib.ptr(wpacked_v2), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(out_ds), B, S, CI,
                                                  int(act_domain), _lib.stream_ptr(dev))
        _lib.check(rc, "conv3x3s2_v2")
        return out, out_ds
    if wpacked.dtype != torch.int16 or wpacked.device != dev or wpacked.numel() != SPLIT_PIECES * 3 * 4 * CI * CO or bias.numel() != CO:
        raise ValueError("conv3x3s2_ds: weights must come from pack_conv3x3s2_ds for this channel count, on the input's device")
    out = torch.empty((B, CO, S, S), dtype=torch.float32, device=dev, memory_format=cl)
    out_ds = torch.empty_like(out, memory_format=cl)
    lib = _lib.load()
    nws = lib.hdn_conv3x3_workspace_bytes(B, S, CI, 2)
    if nws < 0:
        _lib.check(int(nws), "conv3x3s2_ds")
    ws = torch.empty(nws // 4, dtype=torch.float32, device=dev) if nws else None
    with _lib.device_guard(dev):
        rc = lib.hdn_conv3x3s2_ds_f32(_lib.ptr(x), _lib.ptr(wpacked), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(out_ds),
                                      _lib.ptr(ws) if ws is not None else None, nws, B, S, CI, int(act_domain), _lib.stream_ptr(dev))
    _lib.check(rc, "conv3x3s2_ds")
    return out, out_ds


# Batches up to this run the blocks CHAINED (hdn_conv3x3_chain_f32): at the tracker's B = 1 every launch is a dependent step of ~5 us
# and the launches that only add the K slices up were half of the trunk's 70
CHAIN_MAX_BATCH = 16


class LazyAct:
    """An activation of the chained trunk that was never written out: relu(sum of `slices` [z,B,S,S,C] + bias[c] (+ res)), finished by
    the convolution that reads it (or by finish()).  res: None, a channels-last activation [B,C,S,S], or raw slices [zr,B,S,S,C]
    (the downsample branch)."""

    __slots__ = ("slices", "bias", "res")

    def __init__(self, slices, bias, res=None):
        if slices.dim() != 5 or slices.shape[2] != slices.shape[3] or not slices.is_contiguous() or bias.numel() != slices.shape[4]:
            raise ValueError(f"LazyAct: slices [z,B,S,S,C] contiguous with a bias of C elements, got {tuple(slices.shape)}, {tuple(bias.shape)}")
        self.slices, self.bias, self.res = slices, bias, res

    @property
    def shape(self):
        z, B, S, _, C = self.slices.shape
        return (B, C, S, S)

    def res_args(self):
        """(pointer, slice count) of the residual for the C ABI: a channels-last activation counts as one slice."""
        import torch

        from . import _lib

        r = self.res
        if r is None:
            return None, 0
        z, B, S, _, C = self.slices.shape
        if r.dim() == 5:
            ok = tuple(r.shape[1:]) == (B, S, S, C) and r.is_contiguous()
        else:
            ok = tuple(r.shape) == (B, C, S, S) and r.is_contiguous(memory_format=torch.channels_last)
        if not ok or r.dtype != torch.float32 or r.device != self.slices.device:
            raise ValueError(f"LazyAct residual must be a channels-last [B,C,S,S] activation or [z,B,S,S,C] slices matching {tuple(self.slices.shape)}")
        return _lib.ptr(r), (r.shape[0] if r.dim() == 5 else 1)

    def finish(self):
        """The activation itself, channels-last [B,C,S,S] (hdn_conv3x3_finish_f32)."""
        import torch

        from . import _lib

        z, B, S, _, C = self.slices.shape
        dev = self.slices.device
        out = torch.empty((B, C, S, S), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        rp, rz = self.res_args()
        with _lib.device_guard(dev):
            rc = _lib.load().hdn_conv3x3_finish_f32(_lib.ptr(self.slices), z, _lib.ptr(self.bias), rp, rz, _lib.ptr(out), B, S, C, _lib.stream_ptr(dev))
        _lib.check(rc, "conv3x3_finish")
        return out


def chain_conv(x, wpacked, stride=1, want_x=False, act_domain=0):
    """One convolution of the chained trunk (hdn_conv3x3_chain_f32): x a channels-last activation [B,CI,SI,SI] or a LazyAct; returns
    (slices [z,B,S,S,CO], downsample slices or None (stride 2), the finished input as an activation or None (want_x, LazyAct input))."""
    import torch

    from . import _lib

    lazy = isinstance(x, LazyAct)
    B, CI, SI, SI2 = x.shape
    src = x.slices if lazy else x
    dev = _lib.require_device(src)
    if SI != SI2 or SI % stride or (not lazy and not x.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError("chain_conv: square channels-last input")
    S, CO = SI // stride, CI * stride
    T = 4 if stride == 2 else 3
    if wpacked.dtype != torch.int16 or wpacked.device != dev or wpacked.numel() != SPLIT_PIECES * 3 * T * CI * CO:
        raise ValueError("chain_conv: weights must come from pack_conv3x3 / pack_conv3x3s2_ds for this channel count")
    lib = _lib.load()
    z = lib.hdn_conv3x3_chain_slices(B, S, CI, stride)
    if z < 0:
        _lib.check(int(z), "conv3x3_chain")
    out = torch.empty((z, B, S, S, CO), dtype=torch.float32, device=dev)
    out_ds = torch.empty_like(out) if stride == 2 else None
    x_out = torch.empty((B, CI, SI, SI), dtype=torch.float32, device=dev, memory_format=torch.channels_last) if (lazy and want_x) else None
    rp, rz = x.res_args() if lazy else (None, 0)
    with _lib.device_guard(dev):
        rc = lib.hdn_conv3x3_chain_f32(_lib.ptr(src), src.shape[0] if lazy else 0, _lib.ptr(x.bias) if lazy else None, rp, rz,
                                       _lib.ptr(x_out) if x_out is not None else None, _lib.ptr(wpacked), _lib.ptr(out),
                                       _lib.ptr(out_ds) if out_ds is not None else None, B, S, CI, stride, int(act_domain), _lib.stream_ptr(dev))
    _lib.check(rc, "conv3x3_chain")
    return out, out_ds, x_out


class FusedBasicBlock(nn.Module):
    """BasicBlock.forward (backbone/resnet.py:78-94) of the BN-folded trunk with its elementwise tail fused.  Stride-1 3x3
    convolutions of MATRIX_CORE_CHANNELS run, bias / residual / ReLU included, as ONE launch of the split-fp16 matrix-core kernel
    (hdn_conv3x3_bias_relu_f32, channels-last only); every other convolution runs bias-free on MIOpen with `relu(y + b1)` /
    `relu(y + b2 + residual)` as one HIP pass each (hdn_bias_relu_f32).  A folded downsample branch contributes its bias to b2 and
    its raw convolution as the residual.  GPU / eval only."""

    def __init__(self, blk: "BasicBlock", matrix_core: bool = False, act_domain: int = 0):
        super().__init__()
        import torch

        self.act_domain = int(act_domain)     # 1: input, output and residuals are x * 2^-ACT_SCALE_LOG2 in memory (a fully fused trunk's interior)

        for c in (blk.conv1, blk.conv2) + ((blk.downsample,) if blk.downsample is not None else ()):
            if not isinstance(c, nn.Conv2d) or c.bias is None:
                raise ValueError("FusedBasicBlock takes a block whose BatchNorms were folded into biased convolutions")
        self.stride = blk.conv1.stride
        self.w1 = nn.Parameter(blk.conv1.weight.detach().clone(), requires_grad=False)
        self.w2 = nn.Parameter(blk.conv2.weight.detach().clone(), requires_grad=False)
        self.register_buffer("b1", blk.conv1.bias.detach().clone())
        b2 = blk.conv2.bias.detach().clone()
        if blk.downsample is not None:
            self.wd = nn.Parameter(blk.downsample.weight.detach().clone(), requires_grad=False)
            self.ds_stride = blk.downsample.stride
            b2 = b2 + blk.downsample.bias.detach()     # (b2 + bd) once, instead of per element: differs from the unfused sum by rounding
        else:
            self.wd = None
        self.register_buffer("b2", b2)
        if self.act_domain:      # the biases of the scaled domain (exact: a power of two); b1 / b2 stay the real ones
            self.register_buffer("b1d", self.b1 * 2.0 ** -ACT_SCALE_LOG2, persistent=False)
            self.register_buffer("b2d", self.b2 * 2.0 ** -ACT_SCALE_LOG2, persistent=False)
        # packed split-fp16 weights for the matrix-core kernel (stride 1, C -> C only)
        dev = self.w1.device
        cin, cout = self.w1.shape[1], self.w1.shape[0]
        use1 = matrix_core and self.stride == (1, 1) and cin == cout and cout in MATRIX_CORE_CHANNELS
        use2 = matrix_core and cout in MATRIX_CORE_CHANNELS
        use_s2 = (matrix_core and self.stride == (2, 2) and self.wd is not None and cout == 2 * cin and cout in MATRIX_CORE_CHANNELS
                  and self.ds_stride == (2, 2))
        self.register_buffer("p1", pack_conv3x3(self.w1).to(dev) if use1 else None)
        self.register_buffer("p2", pack_conv3x3(self.w2).to(dev) if use2 else None)
        self.register_buffer("p1s2", pack_conv3x3s2_ds(self.w1, self.wd).to(dev) if use_s2 else None)
        # the large-batch packing of the same weights (pack_conv3x3_v2 / pack_conv3x3s2_ds_v2): buffers like p1 / p2 / p1s2, so that
        # .to() / .cuda() move them with the module and no host round trip happens at the first batch of V2_MIN_BATCH or more
        # (which a stream capture could not hold); non-persistent: state_dict stays the reference's
        self.register_buffer("p1v2", pack_conv3x3_v2(self.w1).to(dev) if use1 else None, persistent=False)
        self.register_buffer("p2v2", pack_conv3x3_v2(self.w2).to(dev) if use2 else None, persistent=False)
        self.register_buffer("p1s2v2", pack_conv3x3s2_ds_v2(self.w1, self.wd).to(dev) if use_s2 else None, persistent=False)

    def _packed_v2(self, which, batch):
        """which: 1 / 2 = the block's stride-1 convolutions, "s2" = the stride-2 convolution + downsample branch."""
        if batch < V2_MIN_BATCH or FusedBasicBlock.v2_disabled or (which == "s2" and FusedBasicBlock.v2_s2_disabled):
            return None
        return {1: self.p1v2, 2: self.p2v2, "s2": self.p1s2v2}[which]

    v2_disabled = False        # A/B switch (tests, tools/experiments)
    v2_s2_disabled = False     # ... of the stride-2 stages' large-batch form alone

    def forward(self, x):
        import torch
        import torch.nn.functional as F

        def shape_ok(t):   # the kernel's shapes: square, side tied to the channel count (127-px crops), channels-last
            return t.is_contiguous(memory_format=torch.channels_last) and t.shape[2] == t.shape[3] == _MC_SIDE.get(t.shape[1], -1)

        chained = self._chained(x)
        if chained is not None:
            return chained
        dom = self.act_domain
        b1, b2 = (self.b1d, self.b2d) if dom else (self.b1, self.b2)
        if isinstance(x, LazyAct):
            x = x.finish()
        if (self.p1s2 is not None and x.is_contiguous(memory_format=torch.channels_last)
                and x.shape[2] == x.shape[3] == 2 * _MC_SIDE.get(2 * x.shape[1], -1)):
            y, idt = conv3x3s2_ds(x, self.p1s2, b1, wpacked_v2=self._packed_v2("s2", x.shape[0]), act_domain=dom)   # stride-2 convolution + the downsample branch from one staged input
        else:
            if self.p1 is not None and shape_ok(x):
                y = conv3x3_bias_relu(x, self.p1, b1, wpacked_v2=self._packed_v2(1, x.shape[0]), act_domain=dom)
            else:
                y = bias_relu_(F.conv2d(x, self.w1, None, self.stride, 1), b1)       # (linear + ReLU: the same in either domain, with the domain's bias)
            idt = x if self.wd is None else F.conv2d(x, self.wd, None, self.ds_stride)
        if self.p2 is not None and shape_ok(y) and idt.is_contiguous(memory_format=torch.channels_last):
            return conv3x3_bias_relu(y, self.p2, b2, idt, wpacked_v2=self._packed_v2(2, y.shape[0]), act_domain=dom)
        return bias_relu_(F.conv2d(y, self.w2, None, 1, 1), b2, idt)

    chain_disabled = False     # A/B switch for every block at once (tests, tools/experiments)

    def _chained(self, x):
        """The block as two chained launches (None: not this input).  A LazyAct in: the previous block's output, finished while conv1
        stages it (and written out once, as this block's residual); a LazyAct out."""
        import torch

        lazy = isinstance(x, LazyAct)
        if self.p2 is None or getattr(self, "_hdn_no_chain", False) or FusedBasicBlock.chain_disabled:
            return None
        B, C, S, S2 = x.shape
        if not lazy and not (x.is_cuda and x.dtype == torch.float32 and B <= CHAIN_MAX_BATCH and x.is_contiguous(memory_format=torch.channels_last)):
            return None
        dom = self.act_domain
        b1, b2 = (self.b1d, self.b2d) if dom else (self.b1, self.b2)
        if self.p1s2 is not None and S == S2 == 2 * _MC_SIDE.get(2 * C, -1):
            s1, sd, _ = chain_conv(x, self.p1s2, 2, act_domain=dom)
            s2, _, _ = chain_conv(LazyAct(s1, b1), self.p2, act_domain=dom)
            return LazyAct(s2, b2, sd)
        if self.p1 is not None and self.wd is None and S == S2 == _MC_SIDE.get(C, -1):
            s1, _, idt = chain_conv(x, self.p1, 1, want_x=True, act_domain=dom)
            s2, _, _ = chain_conv(LazyAct(s1, b1), self.p2, act_domain=dom)
            return LazyAct(s2, b2, idt if lazy else x)
        return None


def fold_for_inference(net: HomoResNet, channels_last: bool = True, fused_stem: bool = False, fused_epilogue: bool = False,
                       matrix_core: bool = None) -> nn.Module:
    """A copy of `net` with every eval-mode BatchNorm folded into the preceding convolution (weights scaled in
    float64, rounded once) and, optionally, NHWC weights for MIOpen's channels-last kernels.  Measured on MI355X at
    B=64: 2.92 ms (as-is) -> 2.47 ms (folded) -> 2.11 ms (folded + NHWC); outputs agree with the un-folded CPU
    trunk to ~1.5e-6 relative either way (tools/experiments/exp_trunk.py).  The copy does not track later weight changes.
    fused_stem: replace conv1 / relu / maxpool by FusedStem (GPU only, W <= 128).
    fused_epilogue: replace every BasicBlock by FusedBasicBlock (GPU only): 83 elementwise launches per forward -> 32.
    matrix_core (default: fused_epilogue and channels_last): the stride-1 3x3 convolutions of MATRIX_CORE_CHANNELS as one launch of
    the split-fp16 matrix-core kernel each, epilogue included."""
    import copy

    import torch

    net = copy.deepcopy(net).eval()

    def fuse(conv, bn):
        s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True)
        out = out.to(conv.weight.device)
        out.weight.data = (conv.weight.double() * s.view(-1, 1, 1, 1)).float()
        out.bias.data = (bn.bias.double() - bn.running_mean.double() * s).float()
        return out

    net.conv1, net.bn1 = fuse(net.conv1, net.bn1), nn.Identity()
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in layer:
            blk.conv1, blk.bn1 = fuse(blk.conv1, blk.bn1), nn.Identity()
            blk.conv2, blk.bn2 = fuse(blk.conv2, blk.bn2), nn.Identity()
            if blk.downsample is not None:
                blk.downsample = fuse(blk.downsample[0], blk.downsample[1])
    for p in net.parameters():
        p.requires_grad_(False)
    if channels_last:
        import torch as _t
        net = net.to(memory_format=_t.channels_last)
    mc = bool(channels_last) if matrix_core is None else bool(matrix_core)
    # every stage fused and on the matrix cores: the interior runs in the scaled activation domain (ACT_SCALE_LOG2 above)
    dom = 1 if (fused_epilogue and fused_stem and mc and channels_last and os.environ.get("HDN_TRUNK_SCALED_DOMAIN", "1") not in ("", "0")) else 0
    if fused_epilogue:
        for name in ("layer1", "layer2", "layer3", "layer4"):
            setattr(net, name, nn.Sequential(*[FusedBasicBlock(blk, mc, act_domain=dom) for blk in getattr(net, name)]))
        if channels_last:
            import torch as _t
            net = net.to(memory_format=_t.channels_last)
    if fused_stem:  # conv1 (+ folded bn1) + relu + maxpool in one HIP kernel; the stages behind it stay on MIOpen
        net.conv1 = FusedStem(net.conv1, channels_last, out_domain=dom)
        net.relu, net.maxpool = nn.Identity(), nn.Identity()
    net.act_domain = dom
    return net
