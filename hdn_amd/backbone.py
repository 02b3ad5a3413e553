"""The similarity branch's backbone and necks at inference: PyTorch-ROCm (MIOpen) keeps the convolutions, this module removes the
launches around them.

The reference runs `ResNet-50 (stride 8, dilated) -> AdjustAllLayer` twice per frame (hdn/models/model_builder_e2e_unconstrained_v2.py
:131-158: 255-px search crop, 127-px log-polar crop) as eval-mode `conv -> BatchNorm -> ReLU` chains
(hdn/models/backbone/resnet_atrous.py:60-108 Bottleneck.forward, :186-199 ResNet.forward; hdn/models/neck/neck.py:11-51).  At the
tracker's B = 1 every one of those elementwise modules is a launch of its own: per frame 108 BatchNorm, 95 ReLU and 31 residual-add
launches, 1.1 ms of the 3.4 ms frame (profiles/round4_sequence.txt), none of them doing more than a few microseconds of work.

    optimize_similarity_model(model)      # model.backbone / model.neck / model.neck_lp, in place and reversible (restore_...)

  * every eval-mode BatchNorm is folded into the convolution in front of it (weights scaled in float64, rounded once);
  * a Bottleneck becomes  conv -> hdn_bias_relu_f32 -> conv -> hdn_bias_relu_f32 -> conv -> hdn_bias_relu_f32(+ residual):
    three convolutions + three in-place HIP passes instead of three convolutions + seven library launches; a downsample branch is
    one bias-free convolution whose folded shift rides in the last pass's bias;
  * a neck level is one biased 1x1 convolution.

The modules keep their classes' names and parameters (state_dict unchanged): the object gets a subclass whose forward() uses the
folded copy for CUDA float32 inputs in eval mode and the original forward() for anything else (training, CPU tensors).  The
folded copy follows `load_state_dict` (of the module or of any parent: a post-hook re-folds INTO the existing buffers, so a hipGraph
captured around them sees the new weights too); weights edited in place some other way need optimize_similarity_model again.  A switched
module deep-copies and moves like any other, but pickling the MODULE object (torch.save(model), not state_dict) needs
restore_similarity_model first, because the subclass exists only in this process.
hdn_amd.tracker.DeviceTrackerHomo applies it to the model it is given (HDN_FOLD_BACKBONE=0 keeps the modules as they are).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F


def fold_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """(weight, bias) of the convolution that equals eval-mode bn(conv(x)): w' = w * gamma / sqrt(var + eps), b' = beta - mean * (that
    factor) (+ the convolution's own bias through it), computed in float64 and rounded once."""
    if not isinstance(conv, nn.Conv2d) or not isinstance(bn, nn.BatchNorm2d) or conv.groups != 1:
        raise ValueError("fold_conv_bn takes a dense Conv2d and the BatchNorm2d behind it")
    if bn.running_var is None or bn.running_mean is None:
        raise ValueError("fold_conv_bn needs running statistics (track_running_stats)")
    var, mean = bn.running_var.double(), bn.running_mean.double()
    gamma = bn.weight.double() if bn.weight is not None else torch.ones_like(var)
    beta = bn.bias.double() if bn.bias is not None else torch.zeros_like(var)
    s = gamma / torch.sqrt(var + bn.eps)
    w = (conv.weight.double() * s.view(-1, 1, 1, 1)).to(conv.weight.dtype)
    b0 = conv.bias.double() if conv.bias is not None else torch.zeros_like(var)
    b = (beta + (b0 - mean) * s).to(conv.weight.dtype)
    if conv.weight.is_contiguous(memory_format=torch.channels_last) and not conv.weight.is_contiguous():
        w = w.contiguous(memory_format=torch.channels_last)
    return w.detach(), b.detach()


def _epilogue(y, bias, residual=None):
    """y = relu(y + bias[c] (+ residual)) in place: one HIP pass (hdn_bias_relu_f32) for device tensors; CPU tensors (the CPU tests
    of the folding against the reference's modules) take the same arithmetic in torch ops."""
    if y.is_cuda:
        from .trunk import bias_relu_

        return bias_relu_(y, bias, residual)
    y.add_(bias.view(1, -1, 1, 1))
    if residual is not None:
        y.add_(residual)
    return torch.relu_(y)


class _FoldedConv(nn.Module):
    """A convolution with a BatchNorm folded in.  raw(x): without the shift (the caller's epilogue adds it); act(x): relu(conv + shift);
    forward(x): conv + shift (the necks).  (The 1x1 convolutions as hipBLASLt GEMMs with the shift and the ReLU in the GEMM's epilogue
    were measured and dropped: 2-13 us faster per shape in isolation, but under the frame's hipGraph the library picked 60-120-us
    kernels for some of them: profiles/round4_experiments.txt item 9.)"""

    def __init__(self, conv, bn):
        super().__init__()
        w, b = fold_conv_bn(conv, bn)
        self.register_buffer("weight", w)
        self.register_buffer("bias", b)
        self.stride, self.padding, self.dilation = conv.stride, conv.padding, conv.dilation

    def raw(self, x):
        return F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation)

    def act(self, x):
        return _epilogue(self.raw(x), self.bias)

    def forward(self, x):
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)


def _conv_bn_pair(seq):
    """(conv, bn) of a `Sequential(Conv2d, BatchNorm2d)` (the reference's downsample branches and neck levels), else None."""
    if isinstance(seq, nn.Sequential) and len(seq) == 2 and isinstance(seq[0], nn.Conv2d) and isinstance(seq[1], nn.BatchNorm2d):
        return seq[0], seq[1]
    return None


class FusedBottleneck(nn.Module):
    """Bottleneck.forward (resnet_atrous.py:87-108) of a block whose BatchNorms are folded, with `+ shift -> ReLU` and
    `+ shift + residual -> ReLU` as one in-place pass each.  Takes any module with that block's attributes."""

    def __init__(self, blk):
        super().__init__()
        for name in ("conv1", "bn1", "conv2", "bn2", "conv3", "bn3"):
            if not hasattr(blk, name):
                raise ValueError(f"not a Bottleneck: no {name}")
        self.c1, self.c2, self.c3 = _FoldedConv(blk.conv1, blk.bn1), _FoldedConv(blk.conv2, blk.bn2), _FoldedConv(blk.conv3, blk.bn3)
        down = getattr(blk, "downsample", None)
        if down is None:
            self.cd = None
            self.register_buffer("b3", self.c3.bias.clone())
        else:
            pair = _conv_bn_pair(down)
            if pair is None:
                raise ValueError("Bottleneck.downsample is not Sequential(Conv2d, BatchNorm2d)")
            self.cd = _FoldedConv(*pair)
            self.register_buffer("b3", self.c3.bias + self.cd.bias)      # (b3 + b_downsample) once: differs from the unfused sum by rounding

    def forward(self, x):
        y = self.c2.act(self.c1.act(x))
        idt = x if self.cd is None else self.cd.raw(x)
        return _epilogue(self.c3.raw(y), self.b3, idt)


class FusedAtrousResNet(nn.Module):
    """ResNet.forward (resnet_atrous.py:186-199) with folded BatchNorms and FusedBottleneck blocks: conv1 -> shift + ReLU -> maxpool
    -> layer1..4 -> the `used_layers` selection ([stem, p1, p2, p3, p4]; a single level is returned bare, as the reference does)."""

    def __init__(self, net):
        super().__init__()
        for name in ("conv1", "bn1", "maxpool", "layer1", "layer2", "used_layers"):
            if not hasattr(net, name):
                raise ValueError(f"not the reference's ResNet layout: no {name}")
        self.c1 = _FoldedConv(net.conv1, net.bn1)
        self.maxpool = net.maxpool
        self.layers = nn.ModuleList()
        for name in ("layer1", "layer2", "layer3", "layer4"):
            layer = getattr(net, name, None)
            if isinstance(layer, nn.Sequential):
                self.layers.append(nn.Sequential(*[FusedBottleneck(b) for b in layer]))
            elif layer is None or callable(layer):
                self.layers.append(nn.Identity())       # (`lambda x: x` for an unused stage, resnet_atrous.py:134-141)
            else:
                raise ValueError(f"{name}: expected a Sequential of blocks")
        self.used_layers = list(net.used_layers)

    def forward(self, x):
        x_ = self.c1.act(x)
        p1 = self.layers[0](self.maxpool(x_))
        p2 = self.layers[1](p1)
        p3 = self.layers[2](p2)
        p4 = self.layers[3](p3)
        out = [x_, p1, p2, p3, p4]
        out = [out[i] for i in self.used_layers]
        return out[0] if len(out) == 1 else out


def fold_sequentials(module: nn.Module) -> nn.Module:
    """A deep copy of `module` in which every `Sequential(Conv2d, BatchNorm2d)` is one biased convolution (the necks: AdjustLayer.downsample,
    neck.py:14-17); everything else — the crop of AdjustLayer.forward, the level loop of AdjustAllLayer.forward — stays the module's own."""
    import copy

    m = copy.deepcopy(module)

    def walk(parent):
        for name, child in list(parent.named_children()):
            pair = _conv_bn_pair(child)
            if pair is not None:
                setattr(parent, name, _FoldedConv(*pair))
            else:
                walk(child)

    walk(m)
    return m


# ---------------------------------------------------------------------------------------------------------------- in-place switch
_FUSED, _ORIG_CLASS, _HOOK, _BUILD = "_hdn_fused", "_hdn_orig_class", "_hdn_reload_hook", "_hdn_fused_builder"


def _use_fused(mod, x) -> bool:
    fused = mod.__dict__.get(_FUSED)
    x0 = x[0] if isinstance(x, (list, tuple)) else x
    return (fused is not None and not mod.training and torch.is_tensor(x0) and x0.is_cuda and x0.dtype == torch.float32
            and next(fused.buffers()).device == x0.device)


def _refold(mod, incompatible_keys=None):
    """load_state_dict post-hook: the folded copy is recomputed from the module's new weights and written INTO the existing folded buffers
    (same storage: captured hipGraphs keep pointing at them).  Round-4 ADVICE: without it the folded weights silently stayed the old
    snapshot's while state_dict() reported the new one."""
    fused, build = mod.__dict__.get(_FUSED), mod.__dict__.get(_BUILD)
    if fused is None or build is None:
        return
    with torch.no_grad():
        fresh = build(mod)
        dst = dict(fused.named_buffers())
        for name, buf in fresh.named_buffers():
            dst[name].copy_(buf.to(dst[name].device))


def _attach(mod: nn.Module, fused: nn.Module, build=None):
    """mod(x) -> fused(x) for CUDA float32 inputs in eval mode, the class's own forward otherwise; parameters, buffers and state_dict of
    `mod` are untouched (the folded copy is not a registered sub-module).  build(mod) -> a fresh folded copy (for the reload hook)."""
    base = mod.__dict__.get(_ORIG_CLASS, type(mod))

    def forward(self, *args, **kw):
        if len(args) == 1 and not kw and _use_fused(self, args[0]):
            return self.__dict__[_FUSED](args[0])
        return base.forward(self, *args, **kw)

    object.__setattr__(mod, _FUSED, fused.eval())
    object.__setattr__(mod, _ORIG_CLASS, base)
    object.__setattr__(mod, _BUILD, build)
    object.__setattr__(mod, _HOOK, mod.register_load_state_dict_post_hook(_refold) if build is not None else None)
    mod.__class__ = type(base.__name__, (base,), {"forward": forward, "__module__": base.__module__})


def _detach(mod: nn.Module):
    base = mod.__dict__.get(_ORIG_CLASS)
    if base is not None:
        mod.__class__ = base
        hook = mod.__dict__.pop(_HOOK, None)
        if hook is not None:
            hook.remove()
        mod.__dict__.pop(_BUILD, None)
        del mod.__dict__[_ORIG_CLASS], mod.__dict__[_FUSED]


def _original_view(mod):
    """`mod` with its own class for the duration of a re-fold (the builders read attributes only, but fold_sequentials deep-copies: the
    copy must not carry the switched class, whose folded attributes are plain dictionary entries)."""
    import copy

    m = copy.copy(mod)
    m.__dict__ = {k: v for k, v in mod.__dict__.items() if k not in (_FUSED, _ORIG_CLASS, _HOOK, _BUILD)}
    m.__class__ = mod.__dict__.get(_ORIG_CLASS, type(mod))
    return m


def _build_backbone(mod):
    dev = next(mod.parameters()).device
    return FusedAtrousResNet(_original_view(mod)).to(dev)


def _build_neck(mod):
    return fold_sequentials(_original_view(mod))


def optimize_similarity_model(model, strict: bool = False) -> list:
    """Fold / fuse model.backbone, model.neck, model.neck_lp (the reference's ModelBuilder attributes, model_builder…v2.py:44-60) where
    their structure is the reference's; returns the names that were switched.  strict: raise where the structure is not recognised
    instead of leaving that module as it is.  Eval mode only (BatchNorm statistics are frozen into the weights)."""
    done = []
    with torch.no_grad():
        bb = getattr(model, "backbone", None)
        if isinstance(bb, nn.Module):
            try:
                _detach(bb)
                _attach(bb, FusedAtrousResNet(bb), build=_build_backbone)
                done.append("backbone")
            except ValueError:
                if strict:
                    raise
        for name in ("neck", "neck_lp"):
            nk = getattr(model, name, None)
            if isinstance(nk, nn.Module):
                _detach(nk)
                folded = fold_sequentials(nk)
                if any(isinstance(m, _FoldedConv) for m in folded.modules()):
                    _attach(nk, folded, build=_build_neck)
                    done.append(name)
                elif strict:
                    raise ValueError(f"{name}: no Sequential(Conv2d, BatchNorm2d) found")
    return done


def restore_similarity_model(model):
    for name in ("backbone", "neck", "neck_lp"):
        m = getattr(model, name, None)
        if isinstance(m, nn.Module):
            _detach(m)


def enabled() -> bool:
    return os.environ.get("HDN_FOLD_BACKBONE", "1") not in ("", "0")
