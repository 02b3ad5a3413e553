"""HomoModelBuilder / track_proj: the deep-homography head wired onto the HIP kernels.

    HomoModelBuilder.forward(data) -> dict   <- .../Oneline_DLTv1/models/homo_model_builder.py:115-217
    track_proj(model, data, tmp_mask)        <- hdn/models/model_builder_e2e_unconstrained_v2.py:161-217

Per pair the HIP path is: PreShareFeature(template), PreShareFeature(search) -> [PyTorch-ROCm ResNet-34
+ avgpool + fc -> 8 corner offsets] -> fused DLT + projective warp -> PreShareFeature(warped) -> L1 scores.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import homography as G
from .share_feature import PreShareFeature
from .trunk import resnet34_homo


def avgpool_fc(x, fc, in_domain=0):
    """fc(avgpool(x).flatten(1)) as one launch (hdn_avgpool_fc_f32); x [B,C,H,W] float32, NCHW-contiguous or channels-last; in_domain = 1: x is the
    trunk's scaled-domain output (x_real * 2^-8): the pooled means are multiplied by 2^8 (exact) before the fully connected layer."""
    from . import _lib

    dev = _lib.require_device(x, fc.weight)
    B, C, H, W = x.shape
    nhwc = 0 if x.is_contiguous() else 1
    if nhwc and not x.is_contiguous(memory_format=torch.channels_last):
        x, nhwc = x.contiguous(), 0
    w = fc.weight.detach().contiguous()
    out = torch.empty((B, w.shape[0]), dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        rc = _lib.load().hdn_avgpool_fc_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(fc.bias.detach()) if fc.bias is not None else None, _lib.ptr(out),
                                            B, C, H * W, w.shape[0], nhwc, int(in_domain), _lib.stream_ptr(dev))
    _lib.check(rc, "avgpool_fc")
    return out


def _regress(net, feats):
    fast = getattr(net, "_hdn_fast_trunk", None)
    if fast is not None and not net.training:
        inp = feats.contiguous(memory_format=torch.channels_last) if getattr(net, "_hdn_fast_nhwc", False) else feats
        # the tail of the optimised trunk: AdaptiveAvgPool2d(1) + Linear(512, 8) in one launch — which also takes the trunk's output straight from its
        # scaled activation domain (hdn_amd.trunk.ACT_SCALE_LOG2) when the folded copy runs in it
        tail = (feats.is_cuda and feats.dtype == torch.float32 and isinstance(net.avgpool, nn.AdaptiveAvgPool2d) and net.avgpool.output_size in (1, (1, 1))
                and isinstance(net.fc, nn.Linear) and net.fc.out_features <= 16 and net.fc.weight.dtype == torch.float32)
        dom = int(getattr(fast, "act_domain", 0))
        if tail and dom and hasattr(fast, "forward_scaled"):
            return avgpool_fc(fast.forward_scaled(inp).detach(), net.fc, in_domain=1)
        x = fast(inp)
        if tail:
            return avgpool_fc(x.detach(), net.fc)
    else:
        x = net.backbone(feats)
    x = net.avgpool(x)
    x = x.view(x.size(0), -1)
    return net.fc(x)


def optimize_trunk(net, enable: bool = True, channels_last: bool = False, fused_stem: bool = None, fused_epilogue: bool = None):
    """Attach the BN-folded trunk to any module with a `.backbone` (also the reference's HomoModelBuilder).

    Measured at B=64 on MI355X (tools/experiments/exp_trunk.py, fresh process each): as-is 2.92 ms, folded 2.47 ms; with
    torch.backends.cudnn.benchmark = True (MIOpen find mode, set before the first forward): 2.76 / 2.28 ms, and
    folded + channels_last 2.06 ms.  Without find mode channels_last does not pay (2.96 ms), hence the default."""
    from .trunk import fold_for_inference

    on_gpu = next(net.backbone.parameters()).is_cuda
    if fused_stem is None:  # the fused first stage / block epilogues need the HIP library and weights on a GPU
        fused_stem = on_gpu
    if fused_epilogue is None:
        fused_epilogue = on_gpu
    object.__setattr__(net, "_hdn_fast_trunk", fold_for_inference(net.backbone, channels_last, fused_stem, fused_epilogue) if enable else None)
    # the fused stem reads NCHW and writes the layout the rest of the trunk runs in: no input conversion then
    object.__setattr__(net, "_hdn_fast_nhwc", bool(enable and channels_last and not fused_stem))


def _share(net, x):
    return net.ShareFeature(x)


def _check_data(data):
    for k in ("org_imgs", "input_tensors", "h4p", "patch_indices"):
        if k not in data:
            raise KeyError(f"data['{k}'] missing (homo_model_builder.py:116-119)")
    org, inp = data["org_imgs"], data["input_tensors"]
    if org.dim() != 4 or org.shape[1] != 2 or inp.shape != org.shape:
        raise ValueError(f"org_imgs / input_tensors must both be [B,2,H,W] full patches, got {tuple(org.shape)} / {tuple(inp.shape)}")
    B, _, H, W = org.shape
    if tuple(data["patch_indices"].shape) != (B, H * W):
        raise ValueError("patch_indices must be the [B, H*W] full-patch index (get_img_info.py:88-92)")
    return B, H, W


def homo_stages(net, data, cached_patch_1=None):
    """ShareFeature x2 -> trunk -> fused DLT+warp -> ShareFeature.  Returns the intermediates.

    `cached_patch_1`: ShareFeature(template) is constant for a whole sequence; pass it to skip
    one of the three ShareFeature launches.
    """
    B, H, W = _check_data(data)
    org, inp, h4p = data["org_imgs"], data["input_tensors"], data["h4p"]
    with torch.no_grad():
        if cached_patch_1 is None:
            # one launch for both channels: [B,2,H,W] viewed as 2B single-channel images
            both = _share(net, inp.reshape(B * 2, 1, H, W)).reshape(B, 2, H, W)
            p1, p2 = both[:, :1], both[:, 1:]
            feats = both
        else:
            p1 = cached_patch_1
            p2 = _share(net, inp[:, 1:].contiguous())
            feats = torch.cat((p1, p2), dim=1)
        x = _regress(net, feats)
        H_mat, pred = G.dlt_warp(h4p, x, org[:, :1])
        pf = _share(net, pred)
    return {"x": x, "H_mat": H_mat, "pred_I2": pred, "patch_1": p1, "patch_2": p2, "pred_feat": pf}


def track_proj(net, data, tmp_mask=None, cached_patch_1=None):
    """(H_mat [B,3,3], similarity_norm, similarity_norm_simi) as ModelBuilder.track_proj returns them.

    `net` is the HomoModelBuilder (the reference passes self.hm_net's sub-modules); tmp_mask is accepted and
    unused, as in the reference.
    """
    st = homo_stages(net, data, cached_patch_1)
    # model_builder…:213-216 — sample 0 / channel 0 only, divided by the literal 127*127
    inv = 1.0 / (127 * 127)
    score, score_simi = G.l1_score2(st["patch_2"][0, 0], st["pred_feat"][0, 0], st["patch_1"][0, 0], inv)
    return st["H_mat"], score, score_simi


def track_proj_pair(net, template, search, h4p, patch_1, per_sample: bool = False):
    """track_proj for the tracker's per-frame call, where the two crops are separate tensors and ShareFeature(template) is cached
    (SURVEY §3d): the same stages as homo_stages(..., cached_patch_1) without building the [B,2,H,W] pair first (one concatenation
    and one strided copy per call less - launches that are pure latency at B = 1).
    template / search: [B,1,H,W] contiguous fp32; patch_1 = ShareFeature(template).
    per_sample: the two scores of EVERY sample ([B] each) instead of sample 0's (the reference's `[0][0]`): B independent sequences."""
    if template.shape != search.shape or template.dim() != 4 or template.shape[1] != 1:
        raise ValueError(f"template / search must both be [B,1,H,W], got {tuple(template.shape)} / {tuple(search.shape)}")
    with torch.no_grad():
        p2 = _share(net, search.contiguous())
        x = _regress(net, torch.cat((patch_1, p2), dim=1))
        H_mat, pred = G.dlt_warp(h4p, x, template.contiguous())
        pf = _share(net, pred)
    if per_sample:
        score, score_simi = G.l1_score2_batch(p2, pf, patch_1, 1.0 / (127 * 127))
    else:
        score, score_simi = G.l1_score2(p2[0, 0], pf[0, 0], patch_1[0, 0], 1.0 / (127 * 127))
    return H_mat, score, score_simi


class HomoModelBuilder(nn.Module):
    """Same sub-module names as the reference (ShareFeature, backbone, avgpool, fc) so snapshots load."""

    def __init__(self, pretrained: bool = False):
        super().__init__()
        # `pretrained` fetched ImageNet weights over the network in the reference (backbone/__init__.py:39-49);
        # there is no network here, weights come from the tracker snapshot (hdn/utils/model_load.py).
        self.ShareFeature = PreShareFeature()
        self.backbone = resnet34_homo()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, 8)

    def track_proj(self, data, tmp_mask=None, cached_patch_1=None):
        return track_proj(self, data, tmp_mask, cached_patch_1)

    def optimize_for_inference(self, enable: bool = True, channels_last: bool = False, fused_stem: bool = None, fused_epilogue: bool = None):
        """Build (or drop) the BN-folded copy of the trunk used by eval-mode forwards (§8f rank 4).
        Call it after the weights are loaded and the module is on its device; call again if they change."""
        optimize_trunk(self, enable, channels_last, fused_stem, fused_epilogue)
        return self

    def forward(self, data):
        """Inference-mode forward with the reference's output keys (homo_model_builder.py:212-215).

        if_pos / if_unsup default to ones as in the reference (:125-132; the 'search_windowx' typo at :122 means the window is
        always ones there too).  With per-sample flags (the training set's, unconstrained_v2_dataset.py:310-312) the
        negative-sample branch (:172-205) runs: the triplet term over the samples with if_pos * if_unsup == 1 only, normalised
        by their count, and homo_neg_loss = mean L2 norm of the predicted offsets of the if_pos == 0 samples.  Losses are
        computed without autograd (this package is the inference path).
        """
        st = homo_stages(self, data)
        p1, p2, pf, pred, x = st["patch_1"], st["patch_2"], st["pred_feat"], st["pred_I2"], st["x"]
        B = x.shape[0]
        homo_neg_loss = torch.zeros((), device=x.device)
        with torch.no_grad():
            n_pos = B * p1.shape[2] * p1.shape[3]          # default flags [B,1,127,127] of ones: nonzero() has one row per pixel
            if "if_pos" in data or "if_unsup" in data:
                ones = torch.ones((B, 1, 127, 127), dtype=torch.float32, device=x.device)
                if_pos, if_unsup = data.get("if_pos", ones).to(x.device), data.get("if_unsup", ones).to(x.device)
                neg_ids = if_pos.eq(0).nonzero().squeeze(1)
                pos_ids = (if_pos * if_unsup).eq(1).nonzero().squeeze(1)
                n_pos = pos_ids.shape[0]
                if neg_ids.shape[0] != 0:
                    p1, p2, pf = p1[pos_ids], p2[pos_ids], pf[pos_ids]
                    homo_neg_loss = torch.sum(torch.norm(x[neg_ids, :], p=2, dim=1)) / neg_ids.shape[0]
            # TripletMarginLoss(margin=1, p=1, reduce=False): pairwise L1 over the last dim, eps 1e-6
            d_ap = ((p2 - pf) + 1e-6).abs().sum(-1)
            d_an = ((p2 - p1) + 1e-6).abs().sum(-1)
            loss_mat = (d_ap - d_an + 1.0).clamp_min(0.0)
            feature_loss = (loss_mat.sum() / n_pos / (127 * 127)).reshape(1)
        p2 = st["patch_2"]
        pf = st["pred_feat"]
        return {
            "feature_loss": feature_loss,
            "pred_I2_d": pred[:1],
            "x": x,
            "H_mat": st["H_mat"],
            "patch_2_res_d": p2[:1],
            "pred_I2_CnnFeature_d": pf[:1],
            "homo_neg_loss": homo_neg_loss,
        }
