#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

This script is the only place in the repo that imports /root/reference.  It runs
in the build container only (the GPU box has no /root/reference) and writes
small .npz fixtures = inputs + parameters + the reference's outputs.  Nothing
from the reference's source travels: fixtures are data.

Import recipe: the reference needs yacs / imageio /
memory_profiler / colorama / matplotlib at import time and hard-codes .cuda();
none of that is on the tensor path, so they are replaced by inert stubs and
.cuda() is made a no-op so the reference's own torch-CPU arithmetic runs
verbatim.  cv2 is replaced by tests/golden/cv2_shim.py: the six OpenCV entry
points the tracker loop reaches, served by the oracle's restatements, so that
hdnTrackerHomo.init / track_new themselves execute (gen_tracker_loop).

    python tests/golden/make_golden.py [--reference /root/reference]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = None      # --out: write the fixtures somewhere else (to compare a regeneration with the committed files)
SEED = 20260928


# --------------------------------------------------------------------------- #
# stubs
# --------------------------------------------------------------------------- #
class _CfgNode(dict):
    """Minimal attribute-dict standing in for yacs.config.CfgNode."""

    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = _CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], _CfgNode):
                    self[k] = _CfgNode()
                self[k]._merge(v)
            else:
                self[k] = v

    def merge_from_file(self, path):
        import yaml

        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        pass

    def freeze(self):
        pass

    def defrost(self):
        pass


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    yacs = mod("yacs")
    yacs.config = mod("yacs.config", CfgNode=_CfgNode)
    for name in ("imageio", "colorama", "psutil_stub"):
        mod(name)
    # cv2: the six OpenCV entry points the tracker loop reaches, served by the oracle's restatements (tests/golden/cv2_shim.py);
    # every other attribute raises, so a generator cannot silently run on an unlisted OpenCV call
    root = os.path.dirname(os.path.dirname(HERE))
    for p in (root, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cv2_shim

    sys.modules["cv2"] = cv2_shim
    sys.modules["colorama"].Fore = types.SimpleNamespace(RED="", GREEN="", RESET="", YELLOW="", BLUE="")
    sys.modules["colorama"].Style = types.SimpleNamespace(RESET_ALL="")
    mod("memory_profiler", profile=lambda f: f)
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
            import matplotlib.pyplot  # noqa: F401
        except Exception:
            mpl = mod("matplotlib")
            mpl.pyplot = mod("matplotlib.pyplot")
    # the reference hard-codes .cuda(); keep its arithmetic on the CPU path
    torch.cuda.is_available = lambda: True
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import torch.utils.model_zoo as mz

    mz.load_url = lambda *a, **k: {}
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int


def rng(tag: int):
    return np.random.default_rng(SEED + tag)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrs):
    path = os.path.join(OUT_DIR or HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def relu_normal(g, shape):
    return np.maximum(g.standard_normal(shape, dtype=np.float32), 0.0)


# --------------------------------------------------------------------------- #
# generators
# --------------------------------------------------------------------------- #
def gen_xcorr(ref_xcorr):
    cases = {
        # name: (B, C, Hx, Wx, Hk, Wk)
        "prod_5x29": (2, 16, 29, 29, 5, 5),
        "cfg5_5x35": (1, 8, 35, 35, 5, 5),
        "north_31x61": (1, 4, 61, 61, 31, 31),
        "ragged_3x4_in_7x9": (2, 3, 7, 9, 3, 4),
        "full_size_kernel": (1, 5, 6, 6, 6, 6),
        "one_by_one": (3, 2, 4, 5, 1, 1),
    }
    out = {}
    for i, (name, (B, C, Hx, Wx, Hk, Wk)) in enumerate(cases.items()):
        g = rng(100 + i)
        x = relu_normal(g, (B, C, Hx, Wx))
        k = relu_normal(g, (B, C, Hk, Wk))
        if name == "ragged_3x4_in_7x9":  # signed inputs too
            x = g.standard_normal((B, C, Hx, Wx), dtype=np.float32)
            k = g.standard_normal((B, C, Hk, Wk), dtype=np.float32)
        y = ref_xcorr.xcorr_depthwise(t(x), t(k)).numpy()
        out[name + "__x"] = x
        out[name + "__k"] = k
        out[name + "__y"] = y
    save("xcorr_depthwise", **out)

    # full-channel cases: inputs are re-derived from the seed in the test, outputs are sampled
    samp = {}
    for j, (name, (B, C, Hx, Wx, Hk, Wk)) in enumerate(
        {"prod256_5x29": (1, 256, 29, 29, 5, 5), "north256_31x61": (1, 256, 61, 61, 31, 31)}.items()
    ):
        g = rng(150 + j)
        x = relu_normal(g, (B, C, Hx, Wx))
        k = relu_normal(g, (B, C, Hk, Wk))
        y = ref_xcorr.xcorr_depthwise(t(x), t(k)).numpy()
        idx = rng(160 + j).choice(y.size, size=1024, replace=False)
        samp[name + "__shape"] = np.array([B, C, Hx, Wx, Hk, Wk])
        samp[name + "__idx"] = idx.astype(np.int64)
        samp[name + "__val"] = y.reshape(-1)[idx]
        samp[name + "__sum"] = np.array(y.astype(np.float64).sum())
    save("xcorr_depthwise_sampled", **samp)

    # channel-contracting variants (UPChannelBAN shapes, ban.py:26-47: O = 2 and O = 4)
    fast = {}
    for i, (name, (B, C, O, Hx, Hk)) in enumerate({"cls_o2": (2, 16, 2, 29, 5), "loc_o4": (1, 8, 4, 13, 3)}.items()):
        g = rng(250 + i)
        x = g.standard_normal((B, C, Hx, Hx), dtype=np.float32)
        k = g.standard_normal((B, O * C, Hk, Hk), dtype=np.float32)
        fast[name + "__x"], fast[name + "__k"] = x, k
        fast[name + "__y"] = ref_xcorr.xcorr_fast(t(x), t(k)).numpy()
    # xcorr_slow (xcorr.py:10-23) contracts ALL kernel channels: it only accepts kernel[B,C,h,w] (O = 1)
    g = rng(259)
    xs, ks = g.standard_normal((3, 6, 9, 11), dtype=np.float32), g.standard_normal((3, 6, 4, 3), dtype=np.float32)
    fast["slow__x"], fast["slow__k"] = xs, ks
    fast["slow__y"] = ref_xcorr.xcorr_slow(t(xs), t(ks)).numpy()
    fast["slow__y_fast"] = ref_xcorr.xcorr_fast(t(xs), t(ks)).numpy()
    save("xcorr_fast", **fast)

    cases_c = {
        "prod_13x13": (2, 16, 13, 13, 13, 13),
        "even_8x10_k3x5": (1, 4, 8, 10, 3, 5),
        "odd_7x5_k7x5": (2, 3, 7, 5, 7, 5),
        "small_k_2x2_in_6x6": (1, 2, 6, 6, 2, 2),
    }
    out = {}
    for i, (name, (B, C, Hx, Wx, Hk, Wk)) in enumerate(cases_c.items()):
        g = rng(200 + i)
        x = g.standard_normal((B, C, Hx, Wx), dtype=np.float32)
        k = g.standard_normal((B, C, Hk, Wk), dtype=np.float32)
        y = ref_xcorr.xcorr_depthwise_circular(t(x), t(k)).numpy()
        out[name + "__x"] = x
        out[name + "__k"] = k
        out[name + "__y"] = y
    save("xcorr_depthwise_circular", **out)


def seeded_bn_(bn, g):
    """Non-trivial eval-mode BN statistics so that folding bugs show."""
    n = bn.num_features
    bn.weight.data = t(g.uniform(0.5, 1.5, n).astype(np.float32))
    bn.bias.data = t(g.uniform(-0.3, 0.3, n).astype(np.float32))
    bn.running_mean.data = t(g.uniform(-0.5, 0.5, n).astype(np.float32))
    bn.running_var.data = t(g.uniform(0.5, 2.0, n).astype(np.float32))


def gen_share_feature(ref_pre):
    torch.manual_seed(SEED)
    m = ref_pre.PreShareFeature().eval()
    g = rng(300)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            seeded_bn_(mod, g)
    x = g.standard_normal((2, 1, 127, 127), dtype=np.float32)
    x_small = g.standard_normal((1, 1, 9, 21), dtype=np.float32)
    with torch.no_grad():
        y = m(t(x)).numpy()
        y_small = m(t(x_small)).numpy()
    sd = {k.replace(".", "__"): v.numpy() for k, v in m.state_dict().items()}
    save("share_feature", x=x, y=y, x_small=x_small, y_small=y_small, **{"sd__" + k: v for k, v in sd.items()})
    return m


def gen_dlt(ref_utils):
    g = rng(400)
    src = np.tile(np.array([0, 0, 0, 127, 127, 127, 127, 0], np.float32), (64, 1))
    off = (8.0 * g.standard_normal((64, 8))).astype(np.float32)
    off[0] = 0.0  # identity
    off[1] = np.array([3.25, -1.5] * 4, np.float32)  # pure shift
    H = ref_utils.DLT_solve(t(src), t(off)).numpy()
    # general (non-square) source quadrilaterals, as training uses GT polys (model_builder…:502)
    src2 = src[:16] + (4.0 * g.standard_normal((16, 8))).astype(np.float32)
    off2 = (6.0 * g.standard_normal((16, 8))).astype(np.float32)
    H2 = ref_utils.DLT_solve(t(src2), t(off2)).numpy()
    save("dlt_solve", src=src, off=off, H=H, src2=src2, off2=off2, H2=H2)


def M_mats(B):
    M = torch.tensor([[63.5, 0.0, 63.5], [0.0, 63.5, 63.5], [0.0, 0.0, 1.0]])
    Minv = torch.inverse(M)
    return M.unsqueeze(0).expand(B, 3, 3), Minv.unsqueeze(0).expand(B, 3, 3)


def gen_transform(ref_utils):
    g = rng(500)
    B, Hh, Ww = 6, 127, 127
    img = g.standard_normal((B, 1, Hh, Ww), dtype=np.float32)
    src = np.tile(np.array([0, 0, 0, 127, 127, 127, 127, 0], np.float32), (B, 1))
    off = np.zeros((B, 8), np.float32)
    off[1] = np.array([2.3, -4.7] * 4, np.float32)  # non-integer shift
    off[2] = (3.0 * g.standard_normal(8)).astype(np.float32)  # mild perspective
    off[3] = (8.0 * g.standard_normal(8)).astype(np.float32)
    off[4] = (16.0 * g.standard_normal(8)).astype(np.float32)  # large
    off[5] = (8.0 * g.standard_normal(8)).astype(np.float32)
    H = ref_utils.DLT_solve(t(src), t(off)).squeeze(1)
    # sample 5: a hand-made H with a strong projective row (t varies 0.7..1.3 over the patch)
    Hn = H.clone()
    Hn[5] = torch.tensor([[1.0, 0.02, 1.5], [0.01, 1.0, -2.0], [2.0e-3, -1.0e-3, 1.0]])
    M, Minv = M_mats(B)
    pidx = torch.arange(Hh * Ww, dtype=torch.float32).unsqueeze(0).expand(B, -1)
    base = (torch.arange(B) * Hh * Ww).unsqueeze(1).expand(B, Hh * Ww).reshape(-1)
    y = ref_utils.transform(Hh, Ww, Minv, Hn, M, t(img), pidx, base).numpy()
    save("transform", img=img, H=Hn.numpy(), y=y)

    # transformer() alone on a non-square image, C=1
    img2 = g.standard_normal((2, 1, 20, 33), dtype=np.float32)
    th = torch.tensor(
        [[[1.0, 0.05, 0.02], [-0.03, 0.97, 0.01], [0.01, -0.02, 1.0]], [[0.9, 0.0, 0.1], [0.0, 1.1, -0.1], [0.0, 0.0, 1.0]]]
    )
    y2, cond = ref_utils.transformer(t(img2), th, (20, 33))
    # degenerate denominators: t = x_t exactly (third row [1,0,0]) crosses 0 at the centre column of an
    # odd-width grid, which exercises the reference's `|t| < 1e-7 -> t += 1e-6` nudge (utils.py:237-240)
    img3 = g.standard_normal((1, 1, 15, 17), dtype=np.float32)
    th3 = torch.tensor([[[0.5, 0.0, 0.0], [0.0, 0.5, 0.0], [1.0, 0.0, 0.0]]])
    y3, cond3 = ref_utils.transformer(t(img3), th3, (15, 17))
    save("transformer", img=img2, theta=th.numpy(), y=y2.numpy(), cond=np.array(float(cond)),
         img3=img3, theta3=th3.numpy(), y3=y3.numpy(), cond3=np.array(float(cond3)))


def gen_homo_model(ref_hmb, ref_gi):
    """HomoModelBuilder.forward / track_proj on cfg-1 style inputs (B=2), seeded weights.

    The 85 MB trunk state_dict is NOT committed: the trunk's output x[B,8] is saved
    as an intermediate so the HIP stages after it are pinned, and trunk parity
    (own ResNet34 vs the reference's, same state_dict) is checked in this container
    by tests/test_trunk_vs_reference.py when /root/reference exists.
    """
    torch.manual_seed(SEED + 1)
    m = ref_hmb.HomoModelBuilder(pretrained=False).eval()
    g = rng(600)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            seeded_bn_(mod, g)
    # small regressor head so the offsets are a few px
    m.fc.weight.data = t((0.002 * g.standard_normal((8, 512))).astype(np.float32))
    m.fc.bias.data = t((2.0 * g.standard_normal(8)).astype(np.float32))
    B = 2
    datas = []
    for b in range(B):
        tmp3 = g.integers(0, 256, (127, 127, 3)).astype(np.float64)
        sea3 = g.integers(0, 256, (127, 127, 3)).astype(np.float64)
        mean = np.reshape(np.array([118.93, 113.97, 102.60]), (1, 1, 3))
        std = np.reshape(np.array([69.85, 68.81, 72.45]), (1, 1, 3))
        tmp = np.transpose(np.mean((tmp3 - mean) / std, axis=2, keepdims=True), [2, 0, 1])
        sea = np.transpose(np.mean((sea3 - mean) / std, axis=2, keepdims=True), [2, 0, 1])
        datas.append(ref_gi.merge_tmp_search(tmp, sea))
    data = {
        "org_imgs": torch.stack([torch.Tensor(d["org_imgs"]).float() for d in datas]),
        "input_tensors": torch.stack([torch.Tensor(d["input_tensors"]).float() for d in datas]),
        "patch_indices": torch.stack([torch.Tensor(d["patch_indices"]).float() for d in datas]),
        "h4p": torch.stack([torch.Tensor(d["four_points"]).float() for d in datas]),
    }
    with torch.no_grad():
        out = m(data)
    sf = {("sf__" + k.replace(".", "__")): v.numpy() for k, v in m.ShareFeature.state_dict().items()}
    save(
        "homo_forward",
        org_imgs=data["org_imgs"].numpy(),
        input_tensors=data["input_tensors"].numpy(),
        patch_indices=data["patch_indices"].numpy(),
        h4p=data["h4p"].numpy(),
        x=out["x"].numpy(),
        H_mat=out["H_mat"].numpy(),
        feature_loss=out["feature_loss"].numpy(),
        pred_I2_d=out["pred_I2_d"].numpy(),
        patch_2_res_d=out["patch_2_res_d"].numpy(),
        pred_I2_CnnFeature_d=out["pred_I2_CnnFeature_d"].numpy(),
        homo_neg_loss=np.array(float(out["homo_neg_loss"])),
        fc_w=m.fc.weight.data.numpy(),
        fc_b=m.fc.bias.data.numpy(),
        **sf,
    )
    # the negative-sample branch (homo_model_builder.py:172-205): per-sample if_pos / if_unsup flags as the training set hands
    # them over (hdn/datasets/dataset/unconstrained_v2_dataset.py:310-312,401-404), B = 3 = the two pairs above + their swap
    idx = [0, 1, 0]
    data3 = {k: v[idx].clone() for k, v in data.items()}
    data3["input_tensors"][2] = data3["input_tensors"][2].flip(0)
    data3["org_imgs"][2] = data3["org_imgs"][2].flip(0)
    flags = {"if_pos": torch.tensor([1.0, 0.0, 1.0]), "if_unsup": torch.tensor([1.0, 1.0, 0.0])}
    with torch.no_grad():
        out3 = m({**data3, **flags})
        out3b = m({**data3, "if_pos": torch.tensor([0.0, 1.0, 1.0]), "if_unsup": torch.ones(3)})
    save(
        "homo_forward_neg",
        org_imgs=data3["org_imgs"].numpy(),
        input_tensors=data3["input_tensors"].numpy(),
        patch_indices=data3["patch_indices"].numpy(),
        h4p=data3["h4p"].numpy(),
        if_pos=flags["if_pos"].numpy(),
        if_unsup=flags["if_unsup"].numpy(),
        x=out3["x"].numpy(),
        H_mat=out3["H_mat"].numpy(),
        feature_loss=out3["feature_loss"].numpy(),
        homo_neg_loss=np.array(float(out3["homo_neg_loss"])),
        feature_loss_b=out3b["feature_loss"].numpy(),
        homo_neg_loss_b=np.array(float(out3b["homo_neg_loss"])),
    )
    return m, data


def gen_track_proj(ref_mb, hm_seeded, data):
    """ModelBuilder.track_proj (model_builder_e2e_unconstrained_v2.py:161-217) of the REAL ModelBuilder (83.6 M parameters,
    built under the production YAML) whose hm_net carries the seeded weights of gen_homo_model; same data as
    homo_forward.npz, so only the outputs are stored: (H_mat, similarity_norm, similarity_norm_simi) and the trunk
    output x that the tests inject (the 85 MB trunk is not committed)."""
    torch.manual_seed(SEED + 2)
    mb = ref_mb.ModelBuilder().eval()
    mb.hm_net.load_state_dict(hm_seeded.state_dict(), strict=True)
    mb.hm_net.eval()
    with torch.no_grad():
        H_mat, s, ss = mb.track_proj(data, None)
        # the trunk output the tuple was computed from (same ops as track_proj :181-194)
        p1 = mb.hm_net.ShareFeature(data["input_tensors"][:, :1])
        p2 = mb.hm_net.ShareFeature(data["input_tensors"][:, 1:])
        x = mb.hm_net.fc(mb.hm_net.avgpool(mb.hm_net.backbone(torch.cat((p1, p2), 1))).flatten(1))
        # second tuple with the pair order swapped: sample 0 of the batch is what the scores read ([0][0], :213-216)
        data_sw = {k: v.flip(0).contiguous() for k, v in data.items()}
        H_sw, s_sw, ss_sw = mb.track_proj(data_sw, None)
    save("track_proj", x=x.numpy(), H_mat=H_mat.numpy(), similarity_norm=np.array(float(s), np.float32),
         similarity_norm_simi=np.array(float(ss), np.float32), H_mat_swapped=H_sw.numpy(),
         similarity_norm_swapped=np.array(float(s_sw), np.float32),
         similarity_norm_simi_swapped=np.array(float(ss_sw), np.float32))


def gen_logpolar(ref_lp):
    """STN_Polar.forward (hdn/models/logpolar.py:50-134) = the log-polar resample of track_new_lp."""
    out = {}
    # small case, full tensors: 31x31 crops -> 15x15 log-polar map
    g = rng(700)
    img = (255.0 * g.random((2, 3, 31, 31))).astype(np.float32)
    polar = np.array([[0.0, 0.0], [1.75, -2.5]], np.float32)
    st = ref_lp.STN_Polar(31)
    y, grid = st(t(img), t(polar), [0, 0])
    y2, grid2 = st(t(img), t(polar), [0, 0.3])
    out.update(small_img=img, small_polar=polar, small_y=y.numpy(), small_grid=grid.numpy(), small_y_rot=y2.numpy(),
               small_grid_rot=grid2.numpy())
    # production size: 255x255 -> 127x127; the image is re-derived from the seed in the test, outputs are sampled
    g = rng(701)
    img = (255.0 * g.random((2, 3, 255, 255))).astype(np.float32)
    polar = np.array([[0.0, 0.0], [3.5, -2.25]], np.float32)
    st = ref_lp.STN_Polar(255)
    y, grid = st(t(img), t(polar), [0, 0])
    idx = rng(702).choice(y.numel(), size=4096, replace=False)
    out.update(prod_polar=polar, prod_idx=idx.astype(np.int64), prod_val=y.numpy().reshape(-1)[idx],
               prod_sum=np.array(y.double().sum().item()), prod_grid=grid.numpy()[:, ::9, ::9, :])
    # STN_Polar(255) applied to a 127 x 127 crop, as ModelBuilder.update_template does (model_builder…:98-107): the
    # 127 x 127 grid is still built for image_sz 255, only the normalisation uses the crop's own size
    g = rng(703)
    img = (255.0 * g.random((1, 3, 127, 127))).astype(np.float32)
    polar = np.array([[0.0, 0.0]], np.float32)
    y, grid = st(t(img), t(polar), [0, 0.2])
    idx = rng(704).choice(y.numel(), size=2048, replace=False)
    out.update(tmpl_idx=idx.astype(np.int64), tmpl_val=y.numpy().reshape(-1)[idx], tmpl_sum=np.array(y.double().sum().item()),
               tmpl_grid=grid.numpy()[:, ::9, ::9, :])
    save("logpolar", **out)


def gen_heads(ref_ban, ref_ban_lp):
    """MultiBAN (ban.py:81-127) and MultiCircBAN (ban_lp.py:55-92), 16-channel replicas so the weights fit a fixture."""
    out = {}
    for tag, cls, zsz, xsz in (("ban", ref_ban.MultiBAN, 7, 31), ("circ", ref_ban_lp.MultiCircBAN, 15, 15)):
        torch.manual_seed(SEED + (3 if tag == "ban" else 4))
        m = cls([16, 16, 16], 2, weighted=True).eval()
        g = rng(800 if tag == "ban" else 801)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                seeded_bn_(mod, g)
        m.cls_weight.data = t(g.standard_normal(3).astype(np.float32))
        m.loc_weight.data = t(g.standard_normal(3).astype(np.float32))
        m.loc_scale.data = t(g.uniform(0.5, 1.5, 3).astype(np.float32))
        zfs = [t(g.standard_normal((2, 16, zsz, zsz), dtype=np.float32)) for _ in range(3)]
        xfs = [t(g.standard_normal((2, 16, xsz, xsz), dtype=np.float32)) for _ in range(3)]
        with torch.no_grad():
            c, l = m(zfs, xfs)
        for i in range(3):
            out[f"{tag}__zf{i}"] = zfs[i].numpy()
            out[f"{tag}__xf{i}"] = xfs[i].numpy()
        out[f"{tag}__cls"] = c.numpy()
        out[f"{tag}__loc"] = l.numpy()
        for k, v in m.state_dict().items():
            out[f"{tag}__sd__" + k.replace(".", "__")] = v.numpy()
    save("heads", **out)

    # production width (256 channels, the path fused_forward runs in the tracker): the weights (3.6 M per head) do not
    # fit a fixture, so the modules are re-created from the seed in the test (torch.manual_seed + the same constructor
    # calls = the same init stream; BN statistics / level weights from the numpy generator) and the outputs are stored
    # in full (they are only [1,2,25,25] + [1,2,25,25] and [1,2,13,13] + [1,4,13,13]).
    out = {}
    for tag, cls, zsz, xsz in (("ban", ref_ban.MultiBAN, 7, 31), ("circ", ref_ban_lp.MultiCircBAN, 15, 15)):
        torch.manual_seed(SEED + (13 if tag == "ban" else 14))
        m = cls([256, 256, 256], 2, weighted=True).eval()
        g = rng(810 if tag == "ban" else 811)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                seeded_bn_(mod, g)
        m.cls_weight.data = t(g.standard_normal(3).astype(np.float32))
        m.loc_weight.data = t(g.standard_normal(3).astype(np.float32))
        m.loc_scale.data = t(g.uniform(0.5, 1.5, 3).astype(np.float32))
        zfs = [t(g.standard_normal((1, 256, zsz, zsz), dtype=np.float32)) for _ in range(3)]
        xfs = [t(g.standard_normal((1, 256, xsz, xsz), dtype=np.float32)) for _ in range(3)]
        with torch.no_grad():
            c, l = m(zfs, xfs)
        out[f"{tag}__cls"] = c.numpy()
        out[f"{tag}__loc"] = l.numpy()
        # a checksum of the seeded parameters, so that a drift of torch's init stream is reported as such
        out[f"{tag}__param_sum"] = np.array(sum(float(v.double().sum()) for v in m.state_dict().values()))
    save("heads256", **out)


def gen_frame(ref_bt, ref_gi):
    """The pure-arithmetic part of the tracker's per-frame image handling (no cv2 call is reached):
    SiameseTracker.get_subwindow / get_subwindow_for_homo with original_sz == model_sz (hdn/tracker/base_tracker.py:61-213:
    crop position arithmetic, uint8 channel-mean padding on every side) and get_search_info / get_template_info on 127-px
    crops (get_img_info.py:8-70)."""
    g = rng(900)
    im = g.integers(0, 256, (97, 133, 3)).astype(np.uint8)
    avg = np.mean(im, axis=(0, 1))
    cases = [((66.0, 48.0), 31), ((3.2, 5.7), 31), ((130.4, 95.9), 33), ((66.5, 48.5), 64), ((-4.0, 50.0), 25),
             ((60.0, 110.0), 40), ((66.0, 48.0), 127), ((12.49, 12.51), 24)]
    out = {"im": im, "avg": avg, "pos": np.array([c[0] for c in cases]), "sz": np.array([c[1] for c in cases])}
    for i, (pos, sz) in enumerate(cases):
        a = ref_bt.SiameseTracker.get_subwindow(None, im, np.array(pos), sz, sz, avg)
        b, pts = ref_bt.SiameseTracker.get_subwindow_for_homo(None, im, np.array(pos), sz, sz, avg)
        assert torch.equal(a, b)
        out[f"crop{i}"] = a.numpy().astype(np.uint8)   # values are uint8-valued floats
        out[f"pts{i}"] = np.array(pts, np.float64)
    crop127 = ref_bt.SiameseTracker.get_subwindow(None, im, np.array((66.0, 48.0)), 127, 127, avg)
    s, ps = ref_gi.get_search_info(crop127)
    tt, pt = ref_gi.get_template_info(crop127)
    assert np.array_equal(s, tt)
    out["search_info"] = s          # float64 [1,127,127]
    # similarity homography builder of the tracker (hdn/utils/transform.py:250-298), the H_sim of hdn_tracker_proj_e2e.py:214
    import hdn.utils.transform as ref_tf
    prm = np.array([[320.0, 180.0, 0.0, 1.0, 0.0, 0.0], [320.5, 179.25, 0.12, 1.07, 3.5, -2.25], [100.0, 50.0, -0.4, 0.9, -7.0, 1.5],
                    [10.0, 20.0, 0.0, 1.2, 0.0, 4.0], [64.0, 64.0, 1.3, 1.0, 2.0, 0.0]])
    out["sim_params"] = prm
    out["sim_H"] = np.stack([ref_tf.rot_scale_around_center_shift_tran(*row) for row in prm])
    save("frame", **out)


def _similarity_cases_through_track_new(ref_te, cfg, cases, with_s_x=False):
    """Decode seeded head maps by EXECUTING the reference's hdnTrackerHomo.track_new (hdn_tracker_proj_e2e.py:141-285) around a model that
    returns the case's maps: every value stored — score, pred_c, pscore, best_idx, the 0.05 / 0.25 gates, centre, score_lp,
    pred_center_lp, sim_lp, scale_delta, rot_delta, H_sim — is one of track_new's own local variables, read when it returns
    (_LocalsAtReturn); no statement of the method is re-typed here.  The tracker is initialised on a blank frame with the case's target
    size / position (init :60-120 executed as well); track_proj returns the identity."""

    class _MapsModel(torch.nn.Module):
        def template(self, z):
            pass

        def track_new(self, x, delta=[0, 0]):
            return {"cls": t(self.case[0]), "loc_c": t(self.case[1].copy())}      # (copies: the decode works in place)

        def track_new_lp(self, x, delta=[0, 0]):
            return {"cls_lp": t(self.case[2]), "loc_lp": t(self.case[3].copy())}

        def track_proj(self, data, tmp_mask):
            return torch.eye(3).unsqueeze(0), torch.tensor(0.0), torch.tensor(0.0)

    cfg.CUDA = False
    model = _MapsModel()
    trk = ref_te.hdnTrackerHomo(model)
    frame = np.full((480, 720, 3), 100, np.uint8)
    out = {}
    for n, cs in enumerate(cases):
        size, pos = np.array(cs["size"]), np.array(cs["pos"])
        cfg.TRACK.WINDOW_INFLUENCE = cs["wi"]
        model.case = cs["m"]
        corners = [pos[0] - size[0] / 2, pos[1] - size[1] / 2, pos[0] - size[0] / 2, pos[1] + size[1] / 2,
                   pos[0] + size[0] / 2, pos[1] + size[1] / 2, pos[0] + size[0] / 2, pos[1] - size[1] / 2]
        trk.init(frame, [corners[0], corners[1], size[0], size[1]], [pos[0], pos[1], size[0], size[1], 0.0], corners, corners[:2])
        with torch.no_grad(), _LocalsAtReturn(ref_te.hdnTrackerHomo.track_new.__code__) as cap:
            trk.track_new(1, frame)
        L = cap.locals
        cls, loc_c, cls_lp, loc_lp = cs["m"]
        k = f"c{n}__"
        out.update({k + "cls": cls, k + "loc_c": loc_c, k + "cls_lp": cls_lp, k + "loc_lp": loc_lp, k + "size": size,
                    k + "center_pos": pos, k + "window_influence": np.array(cs["wi"]), k + "init_s_z": np.array(trk.init_s_z),
                    k + "score": L["score"], k + "pred_c": L["pred_c"], k + "pscore": L["pscore"], k + "best_idx": np.array(L["best_idx"]),
                    k + "stop": np.array(L["stop_update_flag"]), k + "center": np.array(L["center"], np.float64),
                    k + "cxcy": np.array([L["cx"], L["cy"]], np.float64), k + "score_lp": L["score_lp"], k + "pred_center_lp": L["pred_center_lp"],
                    k + "best_idx_lp": np.array(L["best_idx_lp"]), k + "sim_lp": np.array(L["sim_lp"], np.float64),
                    k + "best_score": np.array(L["best_score"]), k + "scale_delta": np.array(L["scale_delta"], np.float64),
                    k + "rot_delta": np.array(L["rot_delta"], np.float64), k + "H_sim": L["H_sim"]})
        if with_s_x:
            out[k + "s_x"] = np.array(L["s_x"])
        assert np.array_equal(trk.center_pos, [L["cx"], L["cy"]])
    return out


def gen_similarity(ref_te, cfg):
    """The similarity half of hdnTrackerHomo.track_new (hdn/tracker/hdn_tracker_proj_e2e.py:164-214) from the two heads'
    output maps to H_sim.  Every numeric routine is the reference's own, called on a real hdnTrackerHomo instance (its
    constructor builds the Hanning window :26-29 and the anchor points): hdnTracker._convert_score
    (hdn_tracker.py:84-91), SiameseTracker._convert_c (base_tracker.py:54-59), hdnTracker._convert_logpolar_simi
    (hdn_tracker.py:51-67), rot_scale_around_center_shift_tran (hdn/utils/transform.py:250-298) — and so are the statements
    between them (:172-185 window blend / argmax / 0.05 gate, :203-214 argmax / 0.25 gate / scale_delta / rot_delta): since round 5
    the values come out of the EXECUTED track_new (_similarity_cases_through_track_new; the cv2 calls between them run on
    tests/golden/cv2_shim.py), not out of a re-typed copy of those statements.  cfg.TRACK.WINDOW_INFLUENCE is the production YAML's except where a case
    overrides it (with the shipped value the 0.05 gate cannot fire: the window alone contributes WINDOW_INFLUENCE at the
    centre)."""
    import hdn.utils.transform as ref_tf

    class _NoModel(torch.nn.Module):
        pass

    trk = ref_te.hdnTrackerHomo(_NoModel())
    g = rng(1000)
    wi_prod = float(cfg.TRACK.WINDOW_INFLUENCE)

    def maps(peak=None, peak_lp=None, cls_bias=0.0, lp_bias=0.0, loc_sigma=0.4, lp_sigma=0.3):
        cls = g.standard_normal((1, 2, 25, 25)).astype(np.float32)
        cls[0, 1] += np.float32(cls_bias)
        loc = (loc_sigma * g.standard_normal((1, 2, 25, 25))).astype(np.float32)
        cls_lp = g.standard_normal((1, 2, 13, 13)).astype(np.float32)
        cls_lp[0, 1] += np.float32(lp_bias)
        loc_lp = (lp_sigma * g.standard_normal((1, 4, 13, 13))).astype(np.float32)
        if peak is not None:
            cls[0, 1, peak[0], peak[1]] += np.float32(6.0)
        if peak_lp is not None:
            cls_lp[0, 1, peak_lp[0], peak_lp[1]] += np.float32(6.0)
        return cls, loc, cls_lp, loc_lp

    cases = []
    # 0: clear peaks near the centre; 1: peaks towards the border, larger regression values
    cases.append(dict(m=maps((13, 11), (6, 7)), size=(150.0, 100.0), pos=(320.0, 180.0), wi=wi_prod))
    cases.append(dict(m=maps((4, 20), (2, 10), loc_sigma=1.5, lp_sigma=1.0), size=(201.0, 77.0), pos=(611.5, 402.25), wi=wi_prod))
    # 2: translation gate fires (window influence 0, class-1 logits far below class 0) -> centre (0, 0), sim_lp = [1,1,0,0]
    cases.append(dict(m=maps(None, (5, 5), cls_bias=-8.0), size=(150.0, 100.0), pos=(300.0, 200.0), wi=0.0))
    # 3: log-polar gate fires (every score_lp < 0.25)
    cases.append(dict(m=maps((12, 12), None, lp_bias=-9.0), size=(90.0, 120.0), pos=(100.0, 80.0), wi=wi_prod))
    # 4: exact ties: equal logits at (i, j) and (j, i) (the window is symmetric bit for bit), and two equal log-polar cells
    c4 = maps(None, None)
    for (i, j) in ((9, 15), (15, 9)):
        c4[0][0, 0, i, j], c4[0][0, 1, i, j] = np.float32(-2.0), np.float32(5.0)
    for (i, j) in ((8, 3), (3, 8)):
        c4[2][0, 0, i, j], c4[2][0, 1, i, j] = np.float32(-1.0), np.float32(7.0)
    cases.append(dict(m=c4, size=(150.0, 100.0), pos=(320.0, 180.0), wi=wi_prod))
    # 5: config.py's default window influence (0.45) and a peak the window out-votes
    cases.append(dict(m=maps((1, 1), (11, 2), cls_bias=-1.0), size=(64.0, 64.0), pos=(50.0, 40.0), wi=0.45))
    # 6: no regression at all (zeros) on the centre cell: scale 1 / rotation 0 exactly -> identity branches of H_sim
    c6 = maps((12, 12), (6, 6))
    c6[1][:] = 0
    c6[3][:] = 0
    cases.append(dict(m=c6, size=(150.0, 100.0), pos=(320.0, 180.0), wi=wi_prod))

    out = {"window": trk.window, "points": trk.points, "points_lp": trk.points_lp, "n_cases": np.array(len(cases)),
           "window_influence_production": np.array(wi_prod)}
    out.update(_similarity_cases_through_track_new(ref_te, cfg, cases))
    cfg.TRACK.WINDOW_INFLUENCE = wi_prod
    save("similarity", **out)


def gen_similarity_sigmoid(ref_te, cfg):
    """The decode of gen_similarity under cfg.BAN.KWARGS.cls_out_channels = 1: hdnTracker._convert_score takes its sigmoid branch
    (hdn_tracker.py:85-87) on 1-channel score maps, for both heads.  No shipped YAML selects it; the reference's code has it, so the
    device decode has it too (hdn_similarity_*_f32, cls_channels = 1).  The tracker is built under the switched configuration (its
    constructor reads the key, hdn_tracker_proj_e2e.py:28) and track_new is EXECUTED as in gen_similarity."""
    g = rng(1010)
    wi_prod = float(cfg.TRACK.WINDOW_INFLUENCE)

    def maps(peak=None, peak_lp=None, cls_bias=0.0, lp_bias=0.0, loc_sigma=0.4, lp_sigma=0.3):
        cls = g.standard_normal((1, 1, 25, 25)).astype(np.float32) + np.float32(cls_bias)
        loc = (loc_sigma * g.standard_normal((1, 2, 25, 25))).astype(np.float32)
        cls_lp = g.standard_normal((1, 1, 13, 13)).astype(np.float32) + np.float32(lp_bias)
        loc_lp = (lp_sigma * g.standard_normal((1, 4, 13, 13))).astype(np.float32)
        if peak is not None:
            cls[0, 0, peak[0], peak[1]] += np.float32(6.0)
        if peak_lp is not None:
            cls_lp[0, 0, peak_lp[0], peak_lp[1]] += np.float32(6.0)
        return cls, loc, cls_lp, loc_lp

    cases = []
    # 0 / 1: clear peaks (centre / border); 2: translation gate (window influence 0, sigmoid < 0.05 everywhere); 3: log-polar gate
    # (every sigmoid < 0.25); 4: exact ties on a symmetric pair of cells; 5: saturated logits (sigmoid rounds to 1.0f: the first of them wins)
    cases.append(dict(m=maps((13, 11), (6, 7)), size=(150.0, 100.0), pos=(320.0, 180.0), wi=wi_prod))
    cases.append(dict(m=maps((3, 21), (1, 11), loc_sigma=1.5, lp_sigma=1.0), size=(201.0, 77.0), pos=(611.5, 402.25), wi=wi_prod))
    cases.append(dict(m=maps(None, (5, 5), cls_bias=-9.0), size=(150.0, 100.0), pos=(300.0, 200.0), wi=0.0))
    cases.append(dict(m=maps((12, 12), None, lp_bias=-6.0), size=(90.0, 120.0), pos=(100.0, 80.0), wi=wi_prod))
    c4 = maps(None, None, cls_bias=-1.0, lp_bias=-1.0)
    for (i, j) in ((9, 15), (15, 9)):
        c4[0][0, 0, i, j] = np.float32(5.0)
    for (i, j) in ((8, 3), (3, 8)):
        c4[2][0, 0, i, j] = np.float32(7.0)
    cases.append(dict(m=c4, size=(150.0, 100.0), pos=(320.0, 180.0), wi=wi_prod))
    c5 = maps((12, 12), (6, 6))
    c5[0][0, 0, 12, 12] = np.float32(40.0)
    c5[2][0, 0, 4:7, 6] = np.float32(30.0)
    cases.append(dict(m=c5, size=(150.0, 100.0), pos=(320.0, 180.0), wi=wi_prod))

    prev = int(cfg.BAN.KWARGS.cls_out_channels)
    cfg.BAN.KWARGS.cls_out_channels = 1
    try:
        trk = ref_te.hdnTrackerHomo(torch.nn.Module())
        assert trk.cls_out_channels == 1
        out = {"window": trk.window, "points": trk.points, "points_lp": trk.points_lp, "n_cases": np.array(len(cases)),
               "window_influence_production": np.array(wi_prod)}
        out.update(_similarity_cases_through_track_new(ref_te, cfg, cases))
    finally:
        cfg.BAN.KWARGS.cls_out_channels = prev
        cfg.TRACK.WINDOW_INFLUENCE = wi_prod
    save("similarity_sigmoid", **out)


def gen_config5(ref_te, ref_ban, ref_bt, cfg):
    """BASELINE configs[4] at the tracker level: cfg.TRACK.INSTANCE_SIZE = 303 (score map 31 x 31, hdn_tracker_proj_e2e.py:24-25;
    37 x 37 search features -> conv_search 35 x 35 -> 31 x 31, ban.py:73-78).
      similarity303.npz  the decode of gen_similarity on a REAL hdnTrackerHomo built under INSTANCE_SIZE = 303 (31 x 31 window /
                         anchor points from its constructor); the log-polar maps stay 13 x 13 (cfg.TRAIN.OUTPUT_SIZE_LP)
      heads256_cfg5.npz  the 256-channel MultiBAN on 37 x 37 search features (modules re-created from the seed in the tests)
      frame303.npz       SiameseTracker.get_subwindow with model_sz = original_sz = 303 (crop / pad arithmetic; no cv2 call reached)"""
    import hdn.utils.transform as ref_tf

    class _NoModel(torch.nn.Module):
        pass

    old = cfg.TRACK.INSTANCE_SIZE
    cfg.TRACK.INSTANCE_SIZE = 303
    try:
        trk = ref_te.hdnTrackerHomo(_NoModel())
        assert trk.score_size == 31
        g = rng(1100)
        wi_prod = float(cfg.TRACK.WINDOW_INFLUENCE)
        S = trk.score_size

        def maps(peak=None, peak_lp=None, cls_bias=0.0, lp_bias=0.0, loc_sigma=0.4, lp_sigma=0.3):
            cls = g.standard_normal((1, 2, S, S)).astype(np.float32)
            cls[0, 1] += np.float32(cls_bias)
            loc = (loc_sigma * g.standard_normal((1, 2, S, S))).astype(np.float32)
            cls_lp = g.standard_normal((1, 2, 13, 13)).astype(np.float32)
            cls_lp[0, 1] += np.float32(lp_bias)
            loc_lp = (lp_sigma * g.standard_normal((1, 4, 13, 13))).astype(np.float32)
            if peak is not None:
                cls[0, 1, peak[0], peak[1]] += np.float32(6.0)
            if peak_lp is not None:
                cls_lp[0, 1, peak_lp[0], peak_lp[1]] += np.float32(6.0)
            return cls, loc, cls_lp, loc_lp

        cases = []
        # 0: peaks near the centre; 1: a peak in the outermost ring (a 64-px displacement: what the wider window is for)
        cases.append(dict(m=maps((16, 14), (6, 7)), size=(150.0, 100.0), pos=(320.0, 180.0), wi=wi_prod))
        cases.append(dict(m=maps((1, 29), (2, 10), loc_sigma=1.5, lp_sigma=1.0), size=(201.0, 77.0), pos=(611.5, 402.25), wi=wi_prod))
        # 2: translation gate fires; 3: log-polar gate fires
        cases.append(dict(m=maps(None, (5, 5), cls_bias=-8.0), size=(150.0, 100.0), pos=(300.0, 200.0), wi=0.0))
        cases.append(dict(m=maps((15, 15), None, lp_bias=-9.0), size=(90.0, 120.0), pos=(100.0, 80.0), wi=wi_prod))
        # 4: exact argmax tie at (i, j) / (j, i) (the window is symmetric bit for bit)
        c4 = maps(None, None)
        for (i, j) in ((11, 19), (19, 11)):
            c4[0][0, 0, i, j], c4[0][0, 1, i, j] = np.float32(-2.0), np.float32(5.0)
        cases.append(dict(m=c4, size=(150.0, 100.0), pos=(320.0, 180.0), wi=wi_prod))

        out = {"window": trk.window, "points": trk.points, "points_lp": trk.points_lp, "n_cases": np.array(len(cases)),
               "score_size": np.array(S), "window_influence_production": np.array(wi_prod)}
        out.update(_similarity_cases_through_track_new(ref_te, cfg, cases, with_s_x=True))
        cfg.TRACK.WINDOW_INFLUENCE = wi_prod
        save("similarity303", **out)
    finally:
        cfg.TRACK.INSTANCE_SIZE = old

    # 256-channel MultiBAN on 37 x 37 search features (7 x 7 template): conv_search -> 35 x 35, 5 x 5 kernel -> 31 x 31
    torch.manual_seed(SEED + 15)
    m = ref_ban.MultiBAN([256, 256, 256], 2, weighted=True).eval()
    g = rng(812)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            seeded_bn_(mod, g)
    m.cls_weight.data = t(g.standard_normal(3).astype(np.float32))
    m.loc_weight.data = t(g.standard_normal(3).astype(np.float32))
    m.loc_scale.data = t(g.uniform(0.5, 1.5, 3).astype(np.float32))
    zfs = [t(g.standard_normal((1, 256, 7, 7), dtype=np.float32)) for _ in range(3)]
    xfs = [t(g.standard_normal((1, 256, 37, 37), dtype=np.float32)) for _ in range(3)]
    with torch.no_grad():
        c, l = m(zfs, xfs)
    assert tuple(c.shape) == (1, 2, 31, 31) and tuple(l.shape) == (1, 2, 31, 31)
    save("heads256_cfg5", ban__cls=c.numpy(), ban__loc=l.numpy(),
         ban__param_sum=np.array(sum(float(v.double().sum()) for v in m.state_dict().values())))

    # get_subwindow / get_subwindow_for_homo at model_sz = original_sz = 303: every padding side, odd / fractional centres
    g = rng(901)
    im = g.integers(0, 256, (331, 417, 3)).astype(np.uint8)
    avg = np.mean(im, axis=(0, 1))
    cases = [((208.0, 165.0), 303), ((20.3, 300.7), 303), ((410.5, 10.5), 303)]
    out = {"im": im, "avg": avg, "pos": np.array([c[0] for c in cases]), "sz": np.array([c[1] for c in cases])}
    for i, (pos, sz) in enumerate(cases):
        a = ref_bt.SiameseTracker.get_subwindow(None, im, np.array(pos), sz, sz, avg)
        b, pts = ref_bt.SiameseTracker.get_subwindow_for_homo(None, im, np.array(pos), sz, sz, avg)
        assert torch.equal(a, b)
        out[f"crop{i}"] = a.numpy().astype(np.uint8)
        out[f"pts{i}"] = np.array(pts, np.float64)
    save("frame303", **out)


# --------------------------------------------------------------------------- #
# the tracker loop itself, executed
# --------------------------------------------------------------------------- #
class _LocalsAtReturn:
    """sys.setprofile hook that copies the local variables of ONE function (identified by its code object) at the moment it
    returns: how the generator reads track_new's intermediates (scale_delta, H_sim, H_hm, the crops ...) without touching or
    re-typing a line of it."""

    def __init__(self, code):
        self.code, self.locals = code, None

    def __call__(self, frame, event, arg):
        if event == "return" and frame.f_code is self.code:
            self.locals = dict(frame.f_locals)

    def __enter__(self):
        sys.setprofile(self)
        return self

    def __exit__(self, *a):
        sys.setprofile(None)


def _crc(a):
    import zlib

    return np.array(zlib.crc32(np.ascontiguousarray(a).tobytes()), np.int64)


TRACKER_LOOP_SEQ = dict(n_frames=13, frame_hw=(360, 640), target_wh=(150, 100))   # tools/synth_sequence.make_sequence(seed=SEED, **this)
TRACKER_LOOP_EVENTS = {"gate_frame": 5, "singular_frame": 9}


def _conditioned_model(ref_te, ref_mb, cfg):
    """The reference's real ModelBuilder, seeded and conditioned as gen_tracker_loop's docstring describes, plus the synthetic sequence it was
    calibrated on and the header entries of a tracker-loop fixture.  Shared by gen_tracker_loop (hdnTrackerHomo) and gen_tracker_loop_simi
    (hdnTracker): both loops run around the same model.  -> (mb, frames, corners, init, out)"""
    from tools.synth_sequence import make_sequence

    class _Conditioned(ref_mb.ModelBuilder):
        """The reference's ModelBuilder with a fixed prior on the class-1 logit maps (see gen_tracker_loop) and a switch that raises
        the homography score of one call."""

        score_bias = 0.0
        cls_bias = cls_lp_bias = 0.0      # (gen_tracker_loop_simi lowers one frame's class-1 logits to take the two gates' branches)

        def track_new(self, x, delta=[0, 0]):
            o = super().track_new(x)
            o["cls"] = torch.cat([o["cls"][:, 0:1], o["cls"][:, 1:2] + self.cls_prior[o["cls"].shape[-1]] + self.cls_bias], dim=1)
            return o

        def track_new_lp(self, x, delta=[0, 0]):
            o = super().track_new_lp(x, delta)
            o["cls_lp"] = torch.cat([o["cls_lp"][:, 0:1], o["cls_lp"][:, 1:2] + self.cls_prior_lp + self.cls_lp_bias], dim=1)
            return o

        def track_proj(self, data, tmp_mask):
            H, s, ss = super().track_proj(data, tmp_mask)
            return H, s + self.score_bias, ss

    torch.manual_seed(SEED + 20)
    mb = _Conditioned()
    g = rng(1200)
    for mod in mb.hm_net.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            seeded_bn_(mod, g)
    # PreShareFeature ends in ONE channel behind a ReLU: a seeded BatchNorm can leave it dead (all zeros, both scores 0).  Re-draw its
    # statistics until at least half of a noise image survives.
    probe_img = t(rng(1201).standard_normal((1, 1, 127, 127), dtype=np.float32))
    for attempt in range(32):
        with torch.no_grad():
            alive = float((mb.hm_net.ShareFeature.eval()(probe_img) > 0).float().mean())
        if alive > 0.5:
            break
        gg = rng(1210 + attempt)
        for mod in mb.hm_net.ShareFeature.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                seeded_bn_(mod, gg)
    assert alive > 0.5
    mb.hm_net.fc.weight.data = t((0.01 * g.standard_normal((8, 512))).astype(np.float32))
    fc_bias = t((0.7 * g.standard_normal(8)).astype(np.float32))     # (centred on the first frame's pooled features below)
    mb.head.loc_scale.data = torch.full((3,), 0.5)
    mb.head_lp.loc_scale.data = torch.full((3,), 0.05)
    mb.cls_prior = {}
    for S, c in ((25, (12.25, 11.85)), (31, (15.25, 14.85))):
        yy, xx = torch.meshgrid(torch.arange(float(S)), torch.arange(float(S)), indexing="ij")
        mb.cls_prior[S] = 3.0 * torch.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2) / 8.0).reshape(1, 1, S, S)
    yy, xx = torch.meshgrid(torch.arange(13.0), torch.arange(13.0), indexing="ij")
    mb.cls_prior_lp = 4.0 * torch.exp(-((yy - 6.1) ** 2 + (xx - 5.95) ** 2) / 6.0).reshape(1, 1, 13, 13)

    frames, corners, init = make_sequence(seed=SEED, **TRACKER_LOOP_SEQ)
    cfg.CUDA = False
    out = {"seq__n_frames": np.array(TRACKER_LOOP_SEQ["n_frames"]), "seq__frame_hw": np.array(TRACKER_LOOP_SEQ["frame_hw"]),
           "seq__target_wh": np.array(TRACKER_LOOP_SEQ["target_wh"]), "seq__seed": np.array(SEED),
           "seq__frames_crc": np.array([int(_crc(f)) for f in frames], np.int64),
           "seq__gate_frame": np.array(TRACKER_LOOP_EVENTS["gate_frame"]), "seq__singular_frame": np.array(TRACKER_LOOP_EVENTS["singular_frame"]),
           "seq__bbox": np.array(init["bbox"], np.float64), "seq__poly": np.array(init["poly"], np.float64),
           "seq__gt_points": np.array(init["gt_points"], np.float64), "seq__first_point": np.array(init["first_point"], np.float64),
           "window_influence": np.array(float(cfg.TRACK.WINDOW_INFLUENCE)), "context_amount": np.array(float(cfg.TRACK.CONTEXT_AMOUNT))}
    for k, v in mb.hm_net.ShareFeature.state_dict().items():
        out["sf__" + k.replace(".", "__")] = v.numpy().copy()

    # BatchNorm statistics of the similarity branch: cumulative batch statistics over the first frame's template / search crops
    probe = ref_te.hdnTrackerHomo(mb)
    avg0 = np.mean(frames[0], axis=(0, 1))
    pos0 = np.array(init["poly"][:2])
    size0 = np.array(init["poly"][2:4])
    s_z0 = np.floor(np.sqrt((size0[0] + 0.5 * size0.sum()) * (size0[1] + 0.5 * size0.sum())))
    sim_bns = [m for n, m in mb.named_modules() if isinstance(m, torch.nn.BatchNorm2d) and not n.startswith("hm_net")]
    for m in sim_bns:
        m.reset_running_stats()
        m.momentum = None
    mb.train()
    mb.hm_net.eval()
    with torch.no_grad():
        z = probe.get_subwindow(frames[0], pos0, 127, s_z0, avg0, islog=1)
        x = probe.get_subwindow(frames[0], pos0, 255, 2 * s_z0, avg0)
        ref_mb.ModelBuilder.template(mb, z)
        ref_mb.ModelBuilder.track_new(mb, x)
        ref_mb.ModelBuilder.track_new_lp(mb, x, [0, 0])
    mb.eval()
    # the regressor: x = fc_bias + W (f - f0), f0 = pooled trunk features of the (template, template) pair of frame 0 — corner offsets of
    # about a pixel whose data-dependent part is a few tenths (the pooled features of a seeded trunk are O(100): W f alone is +-50 px)
    with torch.no_grad():
        zc, _ = probe.get_subwindow_for_homo(frames[0], pos0, 127, np.floor(np.sqrt(size0[0] * size0[1])), avg0)
        tmp0, _ = ref_te.get_template_info(zc[:, 0:3])
        pair = torch.Tensor(ref_te.merge_tmp_search(tmp0, tmp0)["input_tensors"]).float().unsqueeze(0)
        p0 = mb.hm_net.ShareFeature(pair[:, :1])
        f0 = mb.hm_net.avgpool(mb.hm_net.backbone(torch.cat((p0, p0), dim=1))).flatten(1)
        mb.hm_net.fc.bias.data = fc_bias - (f0 @ mb.hm_net.fc.weight.data.t())[0]

    return mb, frames, corners, init, out


def gen_tracker_loop(ref_te, ref_mb, cfg):
    """hdnTrackerHomo.init / track_new (hdn/tracker/hdn_tracker_proj_e2e.py:60-120,141-285) EXECUTED, verbatim, around the reference's
    real ModelBuilder (83.6 M parameters, production YAML) over a synthetic sequence (tools/synth_sequence.py) -> tracker_loop.npz.

    What runs is the reference's own code: init's size arithmetic, get_subwindow / get_subwindow_for_homo (crop, padding),
    getPolarImg, _convert_score / _convert_c / _convert_logpolar_simi, the window blend and both gates, img_rot_around_center,
    get_search_info / get_template_info / merge_tmp_search, ModelBuilder.template / track_new / track_new_lp / track_proj, the
    inverse / normalise / compose of the residual, the un-scale / un-shift block, the `> 2.5` gate, the H_total recurrence, the singular
    reset.  The only stand-ins are the six OpenCV entry points (tests/golden/cv2_shim.py -> oracle restatements; parity-unpinned).

    The model is seeded, not trained (no snapshot exists in the image), so it is prepared to behave like a tracker's model:
      * hm_net: BatchNorm statistics / fc as gen_homo_model seeds them (corner offsets of a few pixels);
      * similarity branch: BatchNorm running statistics calibrated on the first frame's crops (default statistics let a random
        50-layer network's activations grow to 1e6); loc_scale 0.5 / 0.05; and a fixed Gaussian prior added to the class-1 logit maps by
        a subclass of the reference's ModelBuilder (an untrained head has no peak; with the prior the argmax sits near the centre and
        moves between neighbouring cells with the data).
    Two events exercise the rarely taken branches: at frame `gate_frame` the homography score returned by track_proj is raised by 10
    (the `homo_score > 2.5` branch, :261-262), before frame `singular_frame` the generator sets tracker.H_total to a singular matrix
    (the reset, :150-153).

    Stored per frame: the four head maps and track_proj's outputs (so that tests can replay the networks), the trunk output x, and what the
    reference computed from them — read from track_new's own local variables when it returns (s_x, best_idx, centre, sim_lp, scale_delta,
    rot_delta, H_sim, H_hm, crop_points, H_hm_comp) and from the tracker object (H_total, center_pos, scale, rot) — plus CRC-32 of the
    stabilised frame, the rotated frame and the three crops (frame 1: the crops themselves).  A second, shorter run under
    cfg.TRACK.INSTANCE_SIZE = 303 (BASELINE configs[4]: 31 x 31 score map; the model keeps its STN_Polar(255), the reference's own
    log-polar branch cannot run at 303) is stored under the prefix `b__`."""
    root = os.path.dirname(os.path.dirname(HERE))
    if root not in sys.path:
        sys.path.insert(0, root)
    import cv2_shim
    from tools.synth_sequence import make_sequence

    mb, frames, corners, init, out = _conditioned_model(ref_te, ref_mb, cfg)

    rec = {}
    fc_out = []
    mb.hm_net.fc.register_forward_hook(lambda m, i, o: fc_out.append(o.detach().numpy().copy()))
    for name in ("template", "track_new", "track_new_lp", "track_proj"):
        def wrap(name=name, orig=getattr(mb, name)):
            def f(*a, **k):
                r = orig(*a, **k)
                # copies: _convert_c / _convert_logpolar_simi decode IN PLACE (for a batch of one, permute(1, 2, 3, 0).contiguous() is
                # a view and .numpy() shares its memory: base_tracker.py:54-59, hdn_tracker.py:51-67)
                keep = {kk: vv.clone() for kk, vv in r.items()} if isinstance(r, dict) else r
                rec[name] = (a, keep)
                return r
            return f
        setattr(mb, name, wrap())

    def run(prefix, instance_size, n_track, events):
        old = cfg.TRACK.INSTANCE_SIZE
        cfg.TRACK.INSTANCE_SIZE = instance_size
        try:
            trk = ref_te.hdnTrackerHomo(mb)
            with torch.no_grad():
                trk.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
            P = prefix
            out.update({P + "instance_size": np.array(instance_size), P + "score_size": np.array(trk.score_size), P + "n_track": np.array(n_track),
                        P + "init__z_crop_crc": _crc(rec["template"][0][0].numpy().astype(np.uint8)), P + "init__z_crop_points": np.array(trk.z_crop_points, np.float64),
                        P + "init__z_crop_sm_crc": _crc(trk.z_crop_sm.numpy().astype(np.uint8)),
                        P + "init__z_crop_points_sm": np.array(trk.z_crop_points_sm, np.float64), P + "init__init_s_z": np.array(trk.init_s_z),
                        P + "init__init_s_z_sm": np.array(trk.init_s_z_sm), P + "init__channel_average": np.array(trk.channel_average),
                        P + "init__center_pos": np.array(trk.center_pos), P + "init__size": np.array(trk.size),
                        P + "init__init_homo_tmp_crc": _crc(np.array(trk.init_homo_tmp)), P + "init__init_points": np.array(trk.init_points),
                        P + "init__poly_shift_l": np.array(trk.poly_shift_l)})
            assert np.array_equal(trk.z_crop.numpy(), rec["template"][0][0].numpy())
            if P == "a__":   # (init does not depend on INSTANCE_SIZE: the second run stores the checksums only)
                out.update({P + "init__z_crop": trk.z_crop.numpy().astype(np.uint8), P + "init__init_homo_tmp": np.array(trk.init_homo_tmp)})
            for i in range(1, n_track + 1):
                mb.score_bias = 10.0 if i == events.get("gate_frame") else 0.0
                if i == events.get("singular_frame"):
                    trk.H_total = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 0]], np.float32)
                del cv2_shim.CALLS[:]
                del fc_out[:]
                with torch.no_grad(), _LocalsAtReturn(ref_te.hdnTrackerHomo.track_new.__code__) as cap:
                    res = trk.track_new(i, frames[i])
                L = cap.locals
                names = [c[0] for c in cv2_shim.CALLS]
                assert names == ["warpPerspective", "resize", "resize", "warpAffine", "resize", "warpAffine", "warpPerspective",
                                 "perspectiveTransform"], names
                k = f"{P}f{i}__"
                H_mat, score, simi = rec["track_proj"][1]
                out.update({
                    k + "cls": rec["track_new"][1]["cls"].numpy(), k + "loc_c": rec["track_new"][1]["loc_c"].numpy(),
                    k + "cls_lp": rec["track_new_lp"][1]["cls_lp"].numpy(), k + "loc_lp": rec["track_new_lp"][1]["loc_lp"].numpy(),
                    k + "x": fc_out[-1], k + "H_mat": H_mat.numpy(), k + "homo_score": np.array(L["homo_score"]), k + "simi_score": np.array(float(simi), np.float32),
                    k + "s_x": np.array(L["s_x"]), k + "scale_z": np.array(L["scale_z"]), k + "best_idx": np.array(L["best_idx"]),
                    k + "pscore_best": np.array(L["pscore"][L["best_idx"]]), k + "stop": np.array(L["stop_update_flag"]),
                    k + "center": np.array([L["delta_cx"], L["delta_cy"]], np.float64), k + "cxcy": np.array([L["cx"], L["cy"]], np.float64),
                    k + "best_idx_lp": np.array(L["best_idx_lp"]), k + "pscore_lp_best": np.array(L["pscore_lp"][L["best_idx_lp"]]),
                    k + "sim_lp": np.array(L["sim_lp"], np.float64), k + "best_score": np.array(L["best_score"]),
                    k + "scale_delta": np.array(L["scale_delta"], np.float64), k + "rot_delta": np.array(L["rot_delta"], np.float64),
                    k + "H_sim": np.array(L["H_sim"]), k + "H_hm": np.array(L["H_hm"]), k + "crop_points": np.array(L["crop_points"], np.float64),
                    k + "H_hm_comp": np.array(L["H_hm_comp"]), k + "H_total": np.array(trk.H_total), k + "center_pos": np.array(trk.center_pos),
                    k + "scale": np.array(trk.scale, np.float64), k + "rot": np.array(trk.rot, np.float64),
                    k + "points": np.array(res["points"]), k + "bbox": np.array(res["bbox"], np.float64),
                    k + "img_crc": _crc(L["img"]), k + "rot_img_crc": _crc(L["rot_img_homo"]),
                    k + "x_crop_crc": _crc(L["x_crop"].numpy().astype(np.uint8)), k + "x_crop_moved_crc": _crc(L["x_crop_moved"].numpy().astype(np.uint8)),
                    k + "x_crop_homo_crc": _crc(L["x_crop_homo"].numpy().astype(np.uint8)),
                    k + "search_crc": _crc(np.ascontiguousarray(rec["track_proj"][0][0]["input_tensors"][0, 1].numpy())),
                })
                assert all(np.array_equal(L[n].numpy(), L[n].numpy().astype(np.uint8)) for n in ("x_crop", "x_crop_moved", "x_crop_homo"))
                if i == 1 and P == "a__":
                    out.update({k + "x_crop": L["x_crop"].numpy().astype(np.uint8), k + "x_crop_moved": L["x_crop_moved"].numpy().astype(np.uint8),
                                k + "x_crop_homo": L["x_crop_homo"].numpy().astype(np.uint8)})
                err = np.sqrt(((res["points"].astype(np.float64) - corners[i]) ** 2).sum() / 4)
                print(f"    {P}frame {i}: best_idx {int(L['best_idx'])} lp {int(L['best_idx_lp'])} centre ({L['delta_cx']:+.3f}, {L['delta_cy']:+.3f}) "
                      f"scale {float(L['scale_delta']):.5f} rot {float(L['rot_delta']):+.5f} homo_score {float(L['homo_score']):.4f} vs ground truth {err:.2f} px")
        finally:
            cfg.TRACK.INSTANCE_SIZE = old

    run("a__", 255, TRACKER_LOOP_SEQ["n_frames"] - 1, TRACKER_LOOP_EVENTS)
    run("b__", 303, 5, {"gate_frame": 3})
    save("tracker_loop", **out)

TRACKER_LOOP_SIMI_EVENTS = {"lp_gate_frame": 4, "stop_frame": 8}


def gen_tracker_loop_simi(ref_te, ref_mb, ref_ht, cfg):
    """hdnTracker.init / track_new / update_template (hdn/tracker/hdn_tracker.py:109-301) EXECUTED, verbatim, around the same conditioned
    ModelBuilder and synthetic sequence as gen_tracker_loop -> tracker_loop_simi.npz.  This is TRACKS['hdnTracker']
    (hdn/tracker/tracker_builder.py:13), the similarity-only tracker: the search window follows the target (center_pos / size recurrences on
    the host), and every frame ends with update_template — the FIRST frame rotated by the accumulated rotation, cropped and pushed through
    ModelBuilder.template again.  OpenCV entry points: tests/golden/cv2_shim.py (parity-unpinned, as for tracker_loop.npz).
    Events: at `lp_gate_frame` the log-polar class-1 logits are lowered by 20 (score_lp < 0.25 -> sim_lp = [1, 1, 0, 0], :245-246), at
    `stop_frame` the translation head's (pscore < 0.05 -> centre frozen, stop flag, :207-209; also the lost_count branch :214-222).
    Stored per frame: the four head maps (replayable), track_new's locals at return, the tracker's recurrence state WITH the dtypes numpy gave it
    (rot / lp_shift[1] turn float32 on the first un-gated frame), the result dictionary, CRC-32 of both search crops, of the rotated first
    frame and of the refreshed template crop (frame 1: the crops themselves)."""
    root = os.path.dirname(os.path.dirname(HERE))
    if root not in sys.path:
        sys.path.insert(0, root)
    import cv2_shim

    mb, frames, corners, init, out = _conditioned_model(ref_te, ref_mb, cfg)
    out["scale_score_thresh"] = np.array(float(cfg.TRACK.SCALE_SCORE_THRESH))
    out["seq__lp_gate_frame"], out["seq__stop_frame"] = np.array(TRACKER_LOOP_SIMI_EVENTS["lp_gate_frame"]), np.array(TRACKER_LOOP_SIMI_EVENTS["stop_frame"])
    out = {k: v for k, v in out.items() if not k.startswith("sf__") and k not in ("seq__gate_frame", "seq__singular_frame")}
    cfg.CUDA = False
    rec = {}
    for name in ("template", "track_new", "track_new_lp"):
        def wrap(name=name, orig=getattr(mb, name)):
            def f(*a, **k):
                r = orig(*a, **k)
                keep = {kk: vv.clone() for kk, vv in r.items()} if isinstance(r, dict) else r
                rec[name] = (a, keep)
                return r
            return f
        setattr(mb, name, wrap())
    rotated = {}
    orig_rot = ref_ht.img_rot_around_center

    def rot_spy(img, cx, cy, w, h, rot):
        r = orig_rot(img, cx, cy, w, h, rot)
        rotated["img"], rotated["rot"] = r, rot
        return r
    ref_ht.img_rot_around_center = rot_spy

    def num(v):      # a recurrence value and the dtype numpy left it in ('i' python int, 'd' python float / float64, 'f' float32)
        kind = "f" if isinstance(v, np.float32) else ("i" if isinstance(v, (int, np.integer)) and not isinstance(v, bool) else "d")
        return np.array(float(v), np.float64), np.array(kind)

    try:
        trk = ref_ht.hdnTracker(mb)
        P = "s__"
        n_track = TRACKER_LOOP_SEQ["n_frames"] - 1
        with torch.no_grad():
            trk.init(frames[0], init["bbox"], init["poly"], np.array([init["first_point"]]))
        z0 = rec["template"][0][0].numpy()
        out.update({P + "instance_size": np.array(int(cfg.TRACK.INSTANCE_SIZE)), P + "score_size": np.array(trk.score_size), P + "n_track": np.array(n_track),
                    P + "init__z_crop_crc": _crc(z0.astype(np.uint8)), P + "init__z_crop": z0.astype(np.uint8), P + "init__init_s_z": np.array(trk.init_s_z),
                    P + "init__channel_average": np.array(trk.channel_average), P + "init__center_pos": np.array(trk.center_pos),
                    P + "init__size": np.array(trk.size), P + "init__poly_shift_l": np.array(int(trk.poly_shift_l)),
                    P + "init__scale_coeff": np.array(float(trk.scale_coeff))})
        assert np.array_equal(z0, z0.astype(np.uint8))
        for i in range(1, n_track + 1):
            mb.cls_lp_bias = -20.0 if i == TRACKER_LOOP_SIMI_EVENTS["lp_gate_frame"] else 0.0
            mb.cls_bias = -20.0 if i == TRACKER_LOOP_SIMI_EVENTS["stop_frame"] else 0.0
            del cv2_shim.CALLS[:]
            with torch.no_grad(), _LocalsAtReturn(ref_ht.hdnTracker.track_new.__code__) as cap:
                res = trk.track_new(i, frames[i], None, None)
            L = cap.locals
            names = [c[0] for c in cv2_shim.CALLS]
            assert names == ["resize", "resize", "warpAffine", "resize", "logPolar"], names
            k = f"{P}f{i}__"
            zc = rec["template"][0][0].numpy()
            rot_v, rot_t = num(trk.rot)
            lp_v, lp_t = num(trk.lp_shift[1])
            out.update({
                k + "cls": rec["track_new"][1]["cls"].numpy(), k + "loc_c": rec["track_new"][1]["loc_c"].numpy(),
                k + "cls_lp": rec["track_new_lp"][1]["cls_lp"].numpy(), k + "loc_lp": rec["track_new_lp"][1]["loc_lp"].numpy(),
                k + "s_x": np.array(L["s_x"]), k + "s_z": np.array(L["s_z"]), k + "scale_z": np.array(L["scale_z"]), k + "best_idx": np.array(L["best_idx"]),
                k + "pscore_best": np.array(L["pscore"][L["best_idx"]]), k + "stop": np.array(L["stop_update_flag"]),
                k + "center": np.array([L["center"][0], L["center"][1]], np.float64), k + "cxcy": np.array([L["cx"], L["cy"]], np.float64),
                k + "best_idx_lp": np.array(L["best_idx_lp"]), k + "score_lp_best": np.array(L["pscore_lp"][L["best_idx_lp"]]),
                k + "sim_lp": np.array(L["sim_lp"], np.float64), k + "width": np.array(float(L["width"])), k + "height": np.array(float(L["height"])),
                k + "center_pos": np.array(trk.center_pos), k + "size": np.array(trk.size), k + "rot": rot_v, k + "rot_kind": rot_t,
                k + "lp_shift1": lp_v, k + "lp_shift1_kind": lp_t, k + "scale": np.array(float(trk.scale)), k + "v": np.array(float(trk.v)),
                k + "lost_count": np.array(int(trk.lost_count)), k + "last_lost": np.array(bool(trk.last_lost)),
                k + "window_scale_factor": np.array(float(trk.window_scale_factor)),
                k + "bbox": np.array(res["bbox"], np.float64), k + "bbox_aligned": np.array(res["bbox_aligned"], np.float64),
                k + "best_score": np.array(res["best_score"]), k + "res_rot": np.array(float(res["rot"])), k + "polygon": np.array(res["polygon"], np.float64),
                k + "x_crop_crc": _crc(L["x_crop"].numpy().astype(np.uint8)), k + "x_crop_moved_crc": _crc(L["x_crop_moved"].numpy().astype(np.uint8)),
                k + "rot_init_img_crc": _crc(rotated["img"]), k + "template_rot": np.array(float(rotated["rot"])), k + "z_crop_crc": _crc(zc.astype(np.uint8)),
            })
            assert np.array_equal(zc, zc.astype(np.uint8))
            if i == 1:
                out.update({k + "x_crop": L["x_crop"].numpy().astype(np.uint8), k + "x_crop_moved": L["x_crop_moved"].numpy().astype(np.uint8),
                            k + "z_crop": zc.astype(np.uint8)})
            c = corners[i]
            gt_c = c.mean(0)
            print(f"    {P}frame {i}: best_idx {int(L['best_idx'])} lp {int(L['best_idx_lp'])} stop {int(L['stop_update_flag'])} centre ({L['center'][0]:+.3f}, {L['center'][1]:+.3f}) "
                  f"sim_lp ({float(L['sim_lp'][0]):.5f}, {float(L['sim_lp'][2]):+.5f}) size ({float(L['width']):.3f}, {float(L['height']):.3f}) rot {float(trk.rot):+.5f} [{rot_t}] "
                  f"centre error vs ground truth {np.hypot(L['cx'] - gt_c[0], L['cy'] - gt_c[1]):.2f} px")
    finally:
        ref_ht.img_rot_around_center = orig_rot
    save("tracker_loop_simi", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--only", default="", help="comma-separated fixture names to regenerate (default: all)")
    ap.add_argument("--out", default="", help="directory to write the .npz files to (default: next to this script)")
    args = ap.parse_args()
    if args.out:
        global OUT_DIR
        OUT_DIR = os.path.abspath(args.out)
        os.makedirs(OUT_DIR, exist_ok=True)
    only = set(x for x in args.only.split(",") if x)
    want = lambda *names: not only or bool(only & set(names))
    if not os.path.isdir(args.reference):
        sys.exit(f"reference tree not found at {args.reference}")
    install_stubs()
    sys.path.insert(0, args.reference)
    torch.set_num_threads(1)

    import hdn.core.xcorr as ref_xcorr
    from hdn.core.config import cfg

    cfg.merge_from_file(os.path.join(args.reference, "experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml"))
    import homo_estimator.Deep_homography.Oneline_DLTv1.utils as ref_utils
    import homo_estimator.Deep_homography.Oneline_DLTv1.preprocess.input_feature_extractor as ref_pre
    import homo_estimator.Deep_homography.Oneline_DLTv1.models.homo_model_builder as ref_hmb
    import homo_estimator.Deep_homography.Oneline_DLTv1.tools.get_img_info as ref_gi

    print("generating golden vectors from", args.reference)
    if want("xcorr_depthwise", "xcorr_depthwise_circular", "xcorr_depthwise_sampled", "xcorr_fast"):
        gen_xcorr(ref_xcorr)
    if want("share_feature"):
        gen_share_feature(ref_pre)
    if want("dlt_solve"):
        gen_dlt(ref_utils)
    if want("transform", "transformer"):
        gen_transform(ref_utils)
    if want("homo_forward", "homo_forward_neg", "track_proj"):
        hm_seeded, hm_data = gen_homo_model(ref_hmb, ref_gi)
        import hdn.models.model_builder_e2e_unconstrained_v2 as ref_mb
        gen_track_proj(ref_mb, hm_seeded, hm_data)
    if want("logpolar"):
        import hdn.models.logpolar as ref_lp
        gen_logpolar(ref_lp)
    if want("heads", "heads256"):
        import hdn.models.head.ban as ref_ban
        import hdn.models.head.ban_lp as ref_ban_lp
        gen_heads(ref_ban, ref_ban_lp)
    if want("frame"):
        import hdn.tracker.base_tracker as ref_bt
        gen_frame(ref_bt, ref_gi)
    if want("similarity"):
        import hdn.tracker.hdn_tracker_proj_e2e as ref_te
        gen_similarity(ref_te, cfg)
    if want("similarity_sigmoid"):
        import hdn.tracker.hdn_tracker_proj_e2e as ref_te
        gen_similarity_sigmoid(ref_te, cfg)
    if want("similarity303", "heads256_cfg5", "frame303"):
        import hdn.models.head.ban as ref_ban
        import hdn.tracker.base_tracker as ref_bt
        import hdn.tracker.hdn_tracker_proj_e2e as ref_te
        gen_config5(ref_te, ref_ban, ref_bt, cfg)
    if want("tracker_loop"):
        import hdn.models.model_builder_e2e_unconstrained_v2 as ref_mb
        import hdn.tracker.hdn_tracker_proj_e2e as ref_te
        gen_tracker_loop(ref_te, ref_mb, cfg)
    if want("tracker_loop_simi"):
        import hdn.models.model_builder_e2e_unconstrained_v2 as ref_mb
        import hdn.tracker.hdn_tracker as ref_ht
        import hdn.tracker.hdn_tracker_proj_e2e as ref_te
        gen_tracker_loop_simi(ref_te, ref_mb, ref_ht, cfg)
    print("torch", torch.__version__, "numpy", np.__version__)


if __name__ == "__main__":
    main()
