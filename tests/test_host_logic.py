"""CPU-only tests: C-ABI surface, host-side logic, error behaviour.  No kernel is launched here."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import hdn_oracle as O

import hdn_amd
from hdn_amd import _lib, dist as hdist, install as hinstall, share_feature as SF, xcorr as X


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "hdn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(hdn_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hdn_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "python binding table out of sync with the header"


def test_library_loads_and_reports_abi():
    lib = _lib.load()
    assert lib.hdn_abi_version() == _lib.ABI_VERSION == 10
    assert lib.hdn_last_xcorr_variant() == b"none"


def test_documented_binding_stub_asserts_the_current_abi_version():
    """INTEGRATION.md's copy-paste ctypes stub checks hdn_abi_version(): the number it asserts must be the header's."""
    import re
    header = open(os.path.join(ROOT, "include", "hdn_hip.h")).read()
    abi = int(re.search(r"#define\s+HDN_ABI_VERSION\s+(\d+)", header).group(1))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = [int(v) for v in re.findall(r"hdn_abi_version\(\)\s*==\s*(\d+)", doc)]
    assert stub and all(v == abi for v in stub), (stub, abi)
    from hdn_amd import _lib
    assert _lib.ABI_VERSION == abi


def test_c_abi_argument_errors_need_no_gpu():
    """Argument validation happens before any launch, so the error codes can be checked on a CPU box."""
    lib = _lib.load()
    one = ctypes.c_void_p(16)
    assert lib.hdn_xcorr_depthwise_f32(None, one, one, 1, 1, 5, 5, 3, 3, None) == -1
    assert lib.hdn_xcorr_depthwise_f32(one, one, ctypes.c_void_p(32), 1, 1, 3, 3, 5, 5, None) == -2  # kernel > search
    assert lib.hdn_xcorr_depthwise_f32(one, one, ctypes.c_void_p(32), 0, 1, 5, 5, 3, 3, None) == -2
    assert lib.hdn_xcorr_depthwise_f32(one, one, one, 1, 1, 5, 5, 3, 3, None) == -4  # out aliases x
    assert lib.hdn_share_feature_f32(one, one, ctypes.c_void_p(32), 1, 0, 5, None) == -2
    assert lib.hdn_share_feature_f32(one, one, one, 1, 5, 5, None) == -4
    assert lib.hdn_dlt_solve_f32(one, one, None, 1, None) == -1
    assert lib.hdn_warp_f32(one, one, ctypes.c_void_p(32), 1, 1, 1, 5, None) == -2  # linspace(-1,1,1)
    assert lib.hdn_dlt_warp_f32(one, one, one, one, ctypes.c_void_p(32), 70000, 5, 5, None) == -3
    assert lib.hdn_dlt_warp_strided_f32(one, one, one, 24, one, ctypes.c_void_p(32), 2, 5, 5, None) == -2      # images closer than H * W
    assert lib.hdn_allgather_offsets(one, one, 4, None, None) == -1 and lib.hdn_allgather_offsets(one, one, 0, one, None) == -2
    assert lib.hdn_rccl_comm_create(ctypes.byref(ctypes.c_void_p()), 2, 2, ctypes.c_char_p(b"\0" * 128)) == -2
    assert lib.hdn_rccl_unique_id(None) == -1 and lib.hdn_rccl_available() in (0, 1)
    assert lib.hdn_bias_relu_f32(one, None, None, 1, 4, 4, 1, None) == -1 and lib.hdn_bias_relu_f32(one, one, None, 1, 0, 4, 1, None) == -2
    assert lib.hdn_bias_relu_f32(one, one, one, 1, 4, 4, 1, None) == -4 and lib.hdn_bias_relu_f32(one, one, None, 65536, 512, 4096, 1, None) == -3
    assert lib.hdn_trunk_stem_mfma_f32(one, None, one, one, 8, 127, 127, 0, None) == -1 and lib.hdn_trunk_stem_mfma_f32(one, one, one, ctypes.c_void_p(32), 0, 127, 127, 0, None) == -2
    assert lib.hdn_trunk_stem_mfma_f32(one, one, one, ctypes.c_void_p(32), 8, 126, 127, 0, None) == -3 and lib.hdn_trunk_stem_mfma_f32(one, one, one, one, 8, 127, 127, 0, None) == -4
    assert lib.hdn_conv3x3s2_v2_f32(one, one, one, ctypes.c_void_p(32), None, 24, 16, 64, 0, None) == -1
    assert lib.hdn_conv3x3s2_v2_f32(one, one, one, ctypes.c_void_p(32), ctypes.c_void_p(48), 0, 16, 64, 0, None) == -2
    assert lib.hdn_conv3x3s2_v2_f32(one, one, one, ctypes.c_void_p(32), ctypes.c_void_p(32), 24, 16, 64, 0, None) == -4   # the two outputs alias
    assert lib.hdn_conv3x3s2_v2_f32(one, one, one, ctypes.c_void_p(32), ctypes.c_void_p(48), 24, 16, 128, 0, None) == -3  # not one of the trunk's stages
    assert lib.hdn_conv3x3_bias_relu_f32(one, one, one, None, None, None, 0, 1, 32, 64, 0, None) == -1
    assert lib.hdn_conv3x3_bias_relu_f32(one, one, one, None, ctypes.c_void_p(32), None, 0, 1, 30, 64, 0, None) == -3      # unsupported (S, C)
    assert lib.hdn_conv3x3_bias_relu_f32(one, one, one, None, ctypes.c_void_p(32), None, 0, 1, 32, 64, 0, None) == -1      # needs a workspace at B = 1
    assert lib.hdn_conv3x3_workspace_bytes(64, 32, 64, 1) == 0 and lib.hdn_conv3x3_workspace_bytes(1, 32, 64, 1) > 0 and lib.hdn_conv3x3_workspace_bytes(1, 5, 7, 1) == -3
    bn, ks = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.hdn_conv3x3_pack_info(4, 512, 1, ctypes.byref(bn), ctypes.byref(ks)) == 0 and bn.value == 64 and ks.value >= 1
    assert lib.hdn_conv3x3_pack_info(16, 64, 2, ctypes.byref(bn), ctypes.byref(ks)) == 0 and lib.hdn_conv3x3_pack_info(16, 64, 3, None, None) == -3
    assert lib.hdn_conv3x3s2_ds_f32(one, one, one, ctypes.c_void_p(32), None, None, 0, 1, 16, 64, 0, None) == -1
    assert lib.hdn_conv3x3s2_ds_f32(one, one, one, ctypes.c_void_p(32), ctypes.c_void_p(32), None, 0, 1, 16, 64, 0, None) == -4
    assert lib.hdn_conv3x3s2_ds_f32(one, one, one, ctypes.c_void_p(32), ctypes.c_void_p(48), None, 0, 64, 16, 96, 0, None) == -3
    # the chained form: slice counts are a pure host computation (>= NCHUNK / 2, so that a launch stages at most two chunks)
    assert lib.hdn_conv3x3_chain_slices(1, 32, 64, 1) == 4 and lib.hdn_conv3x3_chain_slices(1, 4, 512, 1) == 32 and lib.hdn_conv3x3_chain_slices(64, 4, 512, 1) == 16
    assert lib.hdn_conv3x3_chain_slices(64, 32, 64, 1) == 2 and lib.hdn_conv3x3_chain_slices(1, 8, 128, 2) == 8
    assert lib.hdn_conv3x3_chain_slices(0, 32, 64, 1) == -2 and lib.hdn_conv3x3_chain_slices(1, 30, 64, 1) == -3
    two = ctypes.c_void_p(128)
    assert lib.hdn_conv3x3_chain_f32(one, 0, None, None, 0, None, one, None, None, 1, 32, 64, 1, 0, None) == -1
    assert lib.hdn_conv3x3_chain_f32(one, 0, None, None, 0, None, one, two, None, 1, 16, 64, 2, 0, None) == -1          # stride 2 writes two outputs
    assert lib.hdn_conv3x3_chain_f32(one, 4, None, None, 0, None, one, two, None, 1, 32, 64, 1, 0, None) == -1          # slices without their bias
    assert lib.hdn_conv3x3_chain_f32(one, 0, None, None, 0, two, one, ctypes.c_void_p(256), None, 1, 32, 64, 1, 0, None) == -2   # an activation has nothing to write out
    assert lib.hdn_conv3x3_chain_f32(one, 4, one, None, 1, None, one, two, None, 1, 32, 64, 1, 0, None) == -2           # residual count without a residual
    assert lib.hdn_conv3x3_chain_f32(one, 4, one, None, 0, two, one, two, None, 1, 32, 64, 1, 0, None) == -4
    assert lib.hdn_conv3x3_chain_f32(one, 0, None, None, 0, None, one, two, None, 1, 32, 64, 3, 0, None) == -2
    assert lib.hdn_conv3x3_finish_f32(one, 4, None, None, 0, two, 1, 32, 64, None) == -1 and lib.hdn_conv3x3_finish_f32(one, 4, one, None, 0, one, 1, 32, 64, None) == -4
    assert lib.hdn_conv3x3_finish_f32(one, 4, one, None, 2, two, 1, 32, 64, None) == -2 and lib.hdn_conv3x3_finish_f32(one, 4, one, None, 0, two, 1, 32, 66, None) == -2
    assert lib.hdn_avgpool_fc_f32(one, one, None, two, 1, 512, 16, 17, 1, 0, None) == -3 and lib.hdn_avgpool_fc_f32(one, one, None, None, 1, 512, 16, 8, 1, 0, None) == -1
    assert lib.hdn_similarity_translation_f32(one, one, one, one, one, None, 1, 25, 0.16, 8.0, 127.0, 2, None) == -1
    assert lib.hdn_similarity_translation_f32(one, one, one, one, one, one, 1, 0, 0.16, 8.0, 127.0, 2, None) == -2
    assert lib.hdn_similarity_logpolar_f32(one, one, None, one, one, 1, 13, 8.0, 0.03, 0.05, 2, None) == -1
    assert lib.hdn_similarity_logpolar_f32(one, one, one, one, one, 0, 13, 8.0, 0.03, 0.05, 2, None) == -2
    assert lib.hdn_similarity_translation_f32(one, one, one, one, one, one, 1, 25, 0.16, 8.0, 127.0, 3, None) == -2      # cls_out_channels: 1 or 2
    assert lib.hdn_similarity_logpolar_f32(one, one, one, one, one, 1, 13, 8.0, 0.03, 0.05, 0, None) == -2
    with pytest.raises(_lib.HdnHipError, match="ncclResult_t 3"):
        _lib.check(-2003, "x")
    with pytest.raises(ValueError):
        _lib.check(-2, "x")
    with pytest.raises(_lib.HdnHipError):
        _lib.check(-1098, "x")


def test_no_cpu_fallback():
    with pytest.raises(_lib.HdnHipError, match="no CPU fallback"):
        hdn_amd.xcorr_depthwise(torch.zeros(1, 1, 5, 5), torch.zeros(1, 1, 3, 3))
    with pytest.raises(_lib.HdnHipError):
        hdn_amd.DLT_solve(torch.zeros(1, 8), torch.zeros(1, 8))
    m = hdn_amd.PreShareFeature().eval()
    with pytest.raises(_lib.HdnHipError):
        m(torch.zeros(1, 1, 8, 8))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hdn_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_xcorr_output_shapes_and_errors():
    assert X.out_shape((2, 256, 29, 29), (2, 256, 5, 5), False) == (2, 256, 25, 25)
    assert X.out_shape((1, 256, 61, 61), (1, 256, 31, 31), False) == (1, 256, 31, 31)
    assert X.out_shape((1, 256, 35, 35), (1, 256, 5, 5), False) == (1, 256, 31, 31)
    assert X.out_shape((3, 256, 13, 13), (3, 256, 13, 13), True) == (3, 256, 13, 13)
    assert X.out_shape((1, 4, 8, 10), (1, 4, 3, 5), True) == (1, 4, 14, 16)  # even sizes: pad = size//2 per side
    for bad in (((1, 2, 5, 5), (1, 3, 3, 3)), ((1, 2, 3, 3), (1, 2, 5, 5)), ((2, 5, 5), (2, 3, 3)), ((0, 2, 5, 5), (0, 2, 3, 3))):
        with pytest.raises(ValueError):
            X.out_shape(bad[0], bad[1], False)
    # the oracle agrees on the shape rule
    y = O.xcorr_depthwise_circular(torch.zeros(1, 4, 8, 10), torch.zeros(1, 4, 3, 5))
    assert tuple(y.shape) == (1, 4, 14, 16)


def _apply_folded(x, f):
    """Evaluate the folded parameter block with plain torch ops in the kernel's layout (host-side check)."""
    f = f.double()
    w1 = f[0:36].reshape(9, 4).t().reshape(4, 1, 3, 3)
    w2 = f[36:324].reshape(2, 9, 8, 2).permute(2, 0, 3, 1).reshape(8, 4, 3, 3)   # [ci/2][k][co][ci%2] -> [co][ci][k]
    w3 = f[324:396].reshape(4, 9, 2).permute(0, 2, 1).reshape(1, 8, 3, 3)        # [ci/2][k][ci%2]     -> [0][ci][k]
    al, be = f[396:409], f[409:422]
    y = x.double()
    for w, sl in ((w1, slice(0, 4)), (w2, slice(4, 12)), (w3, slice(12, 13))):
        y = torch.nn.functional.conv2d(y, w, padding=1)
        y = torch.relu(y * al[sl].view(1, -1, 1, 1) + be[sl].view(1, -1, 1, 1))
    return y.float()


def test_fold_params_layout_and_bn_folding():
    g = load_golden("share_feature")
    sd = {k[4:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}
    f = SF.fold_params(sd)
    assert f.shape == (SF.N_PARAMS,) and f.dtype == torch.float32
    y = _apply_folded(torch.from_numpy(g["x_small"]), f)
    np.testing.assert_allclose(y.numpy(), g["y_small"], atol=2e-6)
    # BN folding is bit-identical to what PyTorch's CPU eval path uses
    xs = torch.linspace(-3, 3, 101).view(1, 1, 1, -1).repeat(1, 4, 1, 1)
    p = lambda n: sd[f"ShareFeature.1.{n}"]
    ref = torch.nn.functional.batch_norm(xs, p("running_mean"), p("running_var"), p("weight"), p("bias"), False, 0.1, 1e-5)
    got = torch.from_numpy(np.float32(xs.double().numpy() * f[396:400].double().view(1, 4, 1, 1).numpy() + f[409:413].double().view(1, 4, 1, 1).numpy()))
    assert (got == ref).float().mean() > 0.999  # fma(x, alpha, beta): single rounding, emulated in float64


def test_preshare_module_state_dict_is_reference_compatible():
    g = load_golden("share_feature")
    sd = {k[4:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}
    m = hdn_amd.PreShareFeature()
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    # training mode runs the stock layers (PyTorch path), eval mode insists on the GPU kernel
    m.train()
    assert m(torch.zeros(2, 1, 8, 8)).shape == (2, 1, 8, 8)
    # folded cache is invalidated when a parameter changes in place
    m.eval()
    k1 = m._param_key("cpu")
    m.ShareFeature[1].running_mean.add_(1.0)
    assert m._param_key("cpu") != k1


def test_homo_model_builder_names_match_reference_layout():
    m = hdn_amd.HomoModelBuilder()
    keys = set(m.state_dict().keys())
    for k in ("ShareFeature.ShareFeature.0.weight", "backbone.conv1.weight", "backbone.layer1.0.conv1.weight",
              "backbone.layer2.0.downsample.0.weight", "backbone.layer4.2.bn2.running_var", "fc.weight", "fc.bias"):
        assert k in keys, k
    assert m.backbone.conv1.weight.shape == (64, 2, 7, 7)
    n = sum(p.numel() for p in m.parameters())
    assert n == 21_286_062  # = the reference HomoModelBuilder (21.29 M)
    with torch.no_grad():
        assert m.backbone(torch.zeros(1, 2, 127, 127)).shape == (1, 512, 4, 4)


def test_linspace_formula_used_by_the_warp_kernel():
    """grid_coord() in dlt_warp.hip: fma(step, i, -1) below n/2, fma(-step, n-1-i, 1) above; bit-equal to torch."""
    for n in (127, 20, 33, 15, 17, 255, 2):
        ref = torch.linspace(-1.0, 1.0, n).numpy()
        step = np.float32(2.0) / np.float32(n - 1)
        i = np.arange(n)
        lo = np.float32(np.float64(step) * i - 1.0)
        hi = np.float32(1.0 - np.float64(step) * (n - 1 - i))
        np.testing.assert_array_equal(np.where(i < n // 2, lo, hi).astype(np.float32), ref)


def test_shard_range_partitions_every_pair_once():
    for n, w in ((512, 8), (64, 1), (10, 4), (3, 8), (0, 2)):
        spans = [hdist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - s for s, e in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        hdist.shard_range(4, 4, 4)


def test_install_rebinds_the_reference_sites():
    names = ["hdn.core.xcorr", "hdn.models.head.ban", "hdn.models.head.ban_lp",
             "homo_estimator.Deep_homography.Oneline_DLTv1.utils",
             "homo_estimator.Deep_homography.Oneline_DLTv1.models.homo_model_builder",
             "homo_estimator.Deep_homography.Oneline_DLTv1.preprocess",
             "hdn.models.model_builder_e2e_unconstrained_v2", "hdn.models.logpolar"]
    mods = {n: types.ModuleType(n) for n in names}
    sentinel = object()
    for n in names:
        for a in ("xcorr_depthwise", "xcorr_depthwise_circular", "xcorr_fast", "xcorr_slow", "DLT_solve", "transform", "transformer", "Homo_STN", "STN_Polar"):
            setattr(mods[n], a, sentinel)
    mods[names[5]].head = {"PreShareFeature": sentinel}

    class ModelBuilder:
        def track_proj(self, data, tmp_mask):
            return sentinel

    mods[names[6]].ModelBuilder = ModelBuilder
    tb = types.ModuleType("hdn.tracker.tracker_builder")
    tb.TRACKS = {"hdnTracker": sentinel, "hdnTrackerHomoProje2e": sentinel}
    mods["hdn.tracker.tracker_builder"] = tb

    class MultiBAN:
        def forward(self, z_fs, x_fs):
            return sentinel

    class MultiCircBAN:
        pass

    mods["hdn.models.head.ban"].MultiBAN = MultiBAN
    mods["hdn.models.head.ban_lp"].MultiCircBAN = MultiCircBAN
    done = hinstall.install(modules=mods)
    assert len(done) == len(hinstall.REBINDINGS) + 4
    assert tb.TRACKS["hdnTrackerHomoProje2e"] is sentinel          # the tracker is only registered on request
    hinstall.uninstall()
    done = hinstall.install(modules=mods, tracker=True)
    assert len(done) == len(hinstall.REBINDINGS) + 6
    from hdn_amd.tracker import DeviceTrackerHomo
    from hdn_amd.simi_tracker import DeviceTrackerSimi
    assert tb.TRACKS["hdnTrackerHomoProje2e"] is DeviceTrackerHomo and tb.TRACKS["hdnTracker"] is DeviceTrackerSimi     # both registry entries
    assert "forward" in MultiBAN.__dict__ and "forward" in MultiCircBAN.__dict__
    assert mods["hdn.models.head.ban"].xcorr_depthwise is hdn_amd.xcorr_depthwise
    assert mods["hdn.models.head.ban_lp"].xcorr_depthwise_circular is hdn_amd.xcorr_depthwise_circular
    assert mods[names[6]].Homo_STN is hdn_amd.transform and mods[names[6]].DLT_solve is hdn_amd.DLT_solve
    assert mods[names[5]].head["PreShareFeature"] is hdn_amd.PreShareFeature
    assert ModelBuilder.track_proj is hinstall._track_proj_method
    assert mods["hdn.models.logpolar"].STN_Polar is hdn_amd.STN_Polar and mods[names[6]].STN_Polar is hdn_amd.STN_Polar
    # training-mode calls are deferred to the class's own forward (the fused one is inference-only) ...
    mb = MultiBAN()
    mb.training = True
    assert mb.forward([], []) is sentinel
    mc = MultiCircBAN()
    mc.training = True
    with pytest.raises(RuntimeError):
        mc.forward([], [])
    # ... and uninstall() puts every site back
    assert hinstall.uninstall() >= len(done)
    assert mods["hdn.models.head.ban"].xcorr_depthwise is sentinel and mods[names[6]].Homo_STN is sentinel
    assert mods[names[5]].head["PreShareFeature"] is sentinel
    assert MultiBAN().forward([], []) is sentinel and "forward" not in MultiCircBAN.__dict__
    assert "_hdn_orig_forward" not in MultiBAN.__dict__
    assert ModelBuilder().track_proj(None, None) is sentinel
    assert tb.TRACKS["hdnTrackerHomoProje2e"] is sentinel and tb.TRACKS["hdnTracker"] is sentinel
    assert hinstall.uninstall() == 0


def test_template_cache_identity_rules():
    """The template-branch cache of fused_forward is keyed on the template tensors THEMSELVES (strong references),
    their in-place version and the conv_kernel parameter versions — not on ids / addresses, which a freed template
    hands to the next one."""
    from hdn_amd import heads
    torch.manual_seed(0)
    box = heads.DepthwiseBAN(4, 4, 2)
    branches = [box.cls, box.loc]
    z = [torch.zeros(1, 4, 7, 7)]
    c = heads._TemplateCache(z, branches, False, kern=["k"])
    assert c.matches(z, branches, False)
    assert not c.matches([torch.zeros(1, 4, 7, 7)], branches, False)      # a different tensor, equal in every integer
    z[0].add_(1.0)
    assert not c.matches(z, branches, False)                                # same tensor, modified in place
    c = heads._TemplateCache(z, branches, False, kern=["k"])
    box.load_state_dict({k: v.clone() for k, v in box.state_dict().items()})
    assert not c.matches(z, branches, False)                                # weights reloaded after the forward
    c = heads._TemplateCache(z, branches, False, kern=["k"])
    assert c.matches(z, branches, False) and not c.matches(z + [z[0]], branches, False)
    assert c.z_fs[0] is z[0]                                                # the entry pins the template it belongs to


def test_logpolar_tables_match_oracle_and_module_signature():
    from hdn_amd import logpolar as LP
    for size, rot in ((127, 0.0), (15, 0.3)):
        for a, b in zip(LP.tables(size, rot), O.logpolar_tables(size, rot)):
            assert torch.equal(a, b)
    m = hdn_amd.STN_Polar(255)
    assert m._orignal_sz == [127, 127]
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 31, 1), torch.zeros(1, 2))
    # any H x W crop is accepted, as in the reference (update_template feeds the 127-px template to STN_Polar(255));
    # on the CPU both reach the device check: no fallback
    for side in (127, 255):
        with pytest.raises(_lib.HdnHipError):
            m(torch.zeros(1, 3, side, side), torch.zeros(1, 2))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_trunk_matches_reference_trunk_on_cpu():
    """Own ResNet-34 definition vs the reference's, same seeded state_dict (CPU, this container only)."""
    import subprocess
    import sys
    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tests/golden")
import make_golden as mg
mg.install_stubs(); sys.path.insert(0, "/root/reference")
import homo_estimator.Deep_homography.Oneline_DLTv1.backbone.resnet as rr
torch.manual_seed(3)
ref = rr.resnet34(used_layers=[4]).eval()
from hdn_amd.trunk import resnet34_homo
mine = resnet34_homo().eval()
assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
mine.load_state_dict(ref.state_dict(), strict=True)
x = torch.randn(2, 2, 127, 127)
with torch.no_grad():
    a, b = ref(x), mine(x)
assert a.shape == b.shape == (2, 512, 4, 4)
assert torch.equal(a, b), float((a - b).abs().max())
print("TRUNK_OK")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "TRUNK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_install_against_the_real_reference_modules():
    """install(strict=True) finds every rebinding site in the actual reference, and a ModelBuilder built afterwards
    carries hdn_amd.PreShareFeature and the fused track_proj (CPU, this container only; nothing is launched)."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tests/golden")
import make_golden as mg
mg.install_stubs(); sys.path.insert(0, "/root/reference")
from hdn.core.config import cfg
cfg.merge_from_file("/root/reference/experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml")
import hdn_amd, hdn_amd.install as hi
done = hi.install(strict=True, tracker=True)
assert len(done) == len(hi.REBINDINGS) + 8, done
import hdn.models.head.ban as ban, hdn.models.head.ban_lp as ban_lp
assert ban.xcorr_depthwise is hdn_amd.xcorr_depthwise and ban_lp.xcorr_depthwise_circular is hdn_amd.xcorr_depthwise_circular
from hdn.models.model_builder_e2e_unconstrained_v2 import ModelBuilder
m = ModelBuilder()
assert isinstance(m.hm_net.ShareFeature, hdn_amd.PreShareFeature)
assert isinstance(m.logpolar_instance, hdn_amd.STN_Polar)
assert ModelBuilder.track_proj is hi._track_proj_method and ModelBuilder.track_new_lp is hi._track_new_lp_method
# build_tracker(model) (hdn/tracker/tracker_builder.py:18-19) now returns the device-resident loop around the reference's model
from hdn.tracker.tracker_builder import build_tracker, TRACKS
from hdn_amd.tracker import DeviceTrackerHomo
from hdn_amd.similarity import DeviceSimilarity
trk = build_tracker(m)
assert type(trk) is DeviceTrackerHomo and trk.net is m.hm_net and isinstance(trk.similarity, DeviceSimilarity) and trk.similarity.model is m
assert trk.cfg.window_influence == cfg.TRACK.WINDOW_INFLUENCE and trk.cfg.score_size == 25 and not m.training
assert all(hasattr(trk, a) for a in ("init", "track_new"))
# ... and cfg.TRACK.TYPE = 'hdnTracker' (hdn/core/config.py:517, experiments/siamban_r50_l234_pot/config.yaml:43) the similarity-only device loop
from hdn_amd.simi_tracker import DeviceTrackerSimi
assert TRACKS["hdnTracker"] is DeviceTrackerSimi
old = cfg.TRACK.TYPE; cfg.TRACK.TYPE = "hdnTracker"
trk2 = build_tracker(m); cfg.TRACK.TYPE = old
assert type(trk2) is DeviceTrackerSimi and trk2.model is m and trk2.scale_score_thresh == cfg.TRACK.SCALE_SCORE_THRESH and trk2.cfg.score_size == 25
# template() still runs the reference's statements, then drops the heads' cached template features
class _Probe(dict): pass
m.head._hdn_template_cache = "stale"; m.head_lp._hdn_template_cache = "stale"
with torch.no_grad():
    m.template(torch.zeros(1, 6, 127, 127))
assert m.head._hdn_template_cache is None and m.head_lp._hdn_template_cache is None and len(m.zf) == 3 and len(m.zf_lp) == 3
n = hi.uninstall()
assert TRACKS["hdnTrackerHomoProje2e"].__name__ == "hdnTrackerHomo" and TRACKS["hdnTracker"].__name__ == "hdnTracker" and "_hdn_wraps" not in vars(ModelBuilder.template)
hi.install(strict=True, tracker=True)
# the reference's own HomoModelBuilder and ours agree on parameter names, so snapshots load either way
ours = hdn_amd.HomoModelBuilder()
assert list(m.hm_net.state_dict().keys()) == list(ours.state_dict().keys())
ours.load_state_dict(m.hm_net.state_dict(), strict=True)
print("INSTALL_OK")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "INSTALL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_backbone_folding_against_the_real_reference_modules():
    """hdn_amd.backbone on the reference's OWN ResNet-50 (resnet_atrous.py, used_layers [2, 3, 4]) and AdjustAllLayer necks, BatchNorm
    statistics randomised: the folded / fused copies reproduce the modules' eval-mode outputs (CPU arithmetic of the same formulas),
    build_tracker(model) switches all three modules, state_dict keys are unchanged, training mode and restore_similarity_model give
    the original forward back."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tests/golden")
import make_golden as mg
mg.install_stubs(); sys.path.insert(0, "/root/reference")
from hdn.core.config import cfg
cfg.merge_from_file("/root/reference/experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml")
import hdn_amd, hdn_amd.install as hi, hdn_amd.backbone as BB
hi.install(strict=True, tracker=True)
from hdn.models.model_builder_e2e_unconstrained_v2 import ModelBuilder
torch.manual_seed(3)
m = ModelBuilder().eval()
for part in (m.backbone, m.neck, m.neck_lp):
    for mod in part.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.6, 1.6); mod.weight.data.uniform_(0.4, 0.9); mod.bias.data.uniform_(-0.2, 0.2)
keys = list(m.state_dict().keys())
x = torch.randn(1, 3, 127, 127)
with torch.no_grad():
    ref_f = m.backbone(x)
    ref_n, ref_l = m.neck(ref_f), m.neck_lp(ref_f)
    fused = BB.FusedAtrousResNet(m.backbone).eval()
    got_f = fused(x)
    got_n, got_l = BB.fold_sequentials(m.neck).eval()(ref_f), BB.fold_sequentials(m.neck_lp).eval()(ref_f)
assert [tuple(t.shape) for t in got_f] == [(1, 512, 15, 15), (1, 1024, 15, 15), (1, 2048, 15, 15)]
for a, b in list(zip(got_f, ref_f)) + list(zip(got_n, ref_n)) + list(zip(got_l, ref_l)):
    assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6, float((a - b).abs().max())
assert [tuple(t.shape) for t in got_n] == [(1, 256, 7, 7)] * 3 and [tuple(t.shape) for t in got_l] == [(1, 256, 15, 15)] * 3
from hdn.tracker.tracker_builder import build_tracker
trk = build_tracker(m)
assert trk.folded == ["backbone", "neck", "neck_lp"], trk.folded
assert list(m.state_dict().keys()) == keys and type(m.backbone).__name__ == "ResNet"
with torch.no_grad():
    again = m.backbone(x)                      # CPU tensors: the class's own forward
assert all(torch.equal(a, b) for a, b in zip(again, ref_f))
BB.restore_similarity_model(m)
assert "_hdn_fused" not in vars(m.backbone) and type(m.backbone).__module__ == "hdn.models.backbone.resnet_atrous"
print("FOLD_OK")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "FOLD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_production_standin_is_the_reference_architecture():
    """tests/production_standin.py (what the GPU box measures configs[3] with) against the reference's own modules: the stand-in's backbone and
    necks take the reference's state_dict STRICTLY (same parameter names and shapes: resnet_atrous.py:113-199, neck.py:11-51) and then
    compute the same features, at both crop sizes — the production-shaped measurement runs the reference's architecture, not a look-alike."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tests"); sys.path.insert(0, "%s/tests/golden")
import make_golden as mg
mg.install_stubs(); sys.path.insert(0, "/root/reference")
from hdn.core.config import cfg
cfg.merge_from_file("/root/reference/experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml")
from hdn.models.model_builder_e2e_unconstrained_v2 import ModelBuilder
import production_standin as PS
torch.manual_seed(11)
m = ModelBuilder().eval()
for part in (m.backbone, m.neck, m.neck_lp):
    for mod in part.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.6, 1.6); mod.weight.data.uniform_(0.4, 0.9); mod.bias.data.uniform_(-0.2, 0.2)
bb, nk, nl = PS.AtrousResNet50().eval(), PS.Necks(True).eval(), PS.Necks(False).eval()
assert list(bb.state_dict().keys()) == list(m.backbone.state_dict().keys())
bb.load_state_dict(m.backbone.state_dict(), strict=True)
nk.load_state_dict(m.neck.state_dict(), strict=True); nl.load_state_dict(m.neck_lp.state_dict(), strict=True)
assert sum(p.numel() for p in bb.parameters()) == sum(p.numel() for p in m.backbone.parameters())
for side, fs, cut in ((127, 15, 7), (255, 31, 31)):
    x = torch.randn(1, 3, side, side)
    with torch.no_grad():
        rf, sf = m.backbone(x), bb(x)
        assert [tuple(t.shape) for t in sf] == [(1, c, fs, fs) for c in (512, 1024, 2048)]
        pairs = list(zip(sf, rf)) + list(zip(nk(sf), m.neck(rf))) + list(zip(nl(sf), m.neck_lp(rf)))
    assert all(a.shape == b.shape and torch.equal(a, b) for a, b in pairs)
    assert tuple(nk(sf)[0].shape) == (1, 256, cut, cut)
print("STANDIN_OK")
''' % (ROOT, ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert "STANDIN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_backbone_switch_keeps_the_module_intact():
    """optimize_similarity_model on a small model of the reference's layout (CPU): class name, parameters and state_dict untouched, CPU / training
    inputs still take the class's own forward, the folded copy reproduces it (torch-op epilogue), deep copies carry their own folded copy,
    restore_similarity_model undoes everything, a second call replaces the first, an unrecognised backbone is left alone (strict: raises)."""
    import copy
    import types
    import torch
    import torch.nn as nn
    from hdn_amd import backbone as BB

    class Block(nn.Module):
        def __init__(self, c, down):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(c, 4, 1, bias=False), nn.BatchNorm2d(4)
            self.conv2, self.bn2 = nn.Conv2d(4, 4, 3, padding=2, dilation=2, bias=False), nn.BatchNorm2d(4)
            self.conv3, self.bn3 = nn.Conv2d(4, 16, 1, bias=False), nn.BatchNorm2d(16)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = nn.Sequential(nn.Conv2d(c, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16)) if down else None

        def forward(self, x):
            y = self.relu(self.bn1(self.conv1(x)))
            y = self.relu(self.bn2(self.conv2(y)))
            y = self.bn3(self.conv3(y))
            return self.relu(y + (x if self.downsample is None else self.downsample(x)))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1, self.relu, self.maxpool = nn.Conv2d(3, 8, 7, 2, 0, bias=False), nn.BatchNorm2d(8), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)
            self.layer1 = nn.Sequential(Block(8, True), Block(16, False))
            self.layer2 = nn.Sequential(Block(16, True))
            self.layer3 = lambda x: x            # (an unused stage, as resnet_atrous.py:134-141 writes it)
            self.used_layers = [0, 1, 2]

        def forward(self, x):
            x_ = self.relu(self.bn1(self.conv1(x)))
            p1 = self.layer1(self.maxpool(x_))
            return [x_, p1, self.layer2(p1)]

    class Neck(nn.Module):
        def __init__(self):
            super().__init__()
            self.downsample = nn.Sequential(nn.Conv2d(16, 5, 1, bias=False), nn.BatchNorm2d(5))

        def forward(self, f):
            return self.downsample(f[2])[:, :, 1:4, 1:4]

    torch.manual_seed(1)
    model = types.SimpleNamespace(backbone=Net().eval(), neck=Neck().eval(), neck_lp=nn.Identity())
    for m in list(model.backbone.modules()) + list(model.neck.modules()):
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.uniform_(-0.5, 0.5); m.running_var.uniform_(0.5, 2); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.5, 0.5)
    keys = list(model.backbone.state_dict().keys())
    x = torch.randn(2, 3, 41, 41)
    with torch.no_grad():
        ref = model.backbone(x)
        ref_n = model.neck(ref)
        assert BB.optimize_similarity_model(model) == ["backbone", "neck"]          # (neck_lp: nothing to fold)
        with pytest.raises(ValueError):
            BB.optimize_similarity_model(model, strict=True)                         # strict: the Identity neck is reported
        bb = model.backbone
        assert type(bb).__name__ == "Net" and isinstance(bb, Net) and list(bb.state_dict().keys()) == keys
        assert all(torch.equal(a, b) for a, b in zip(bb(x), ref))                    # CPU input: the class's own forward
        fused = vars(bb)["_hdn_fused"]
        for a, b in list(zip(fused(x), ref)) + [(vars(model.neck)["_hdn_fused"](ref), ref_n)]:
            assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
        twin = copy.deepcopy(bb)
        assert vars(twin)["_hdn_fused"] is not fused and list(twin.state_dict().keys()) == keys and all(torch.equal(a, b) for a, b in zip(twin(x), ref))
        assert BB.optimize_similarity_model(model) == ["backbone", "neck"] and vars(bb)["_hdn_fused"] is not fused      # again: replaced, not stacked
        assert type(bb).__mro__[1] is Net
        # another snapshot through load_state_dict (round-4 ADVICE: the folded copy used to stay the old snapshot's, silently): the
        # post-hook re-folds into the SAME buffers (a captured hipGraph keeps pointing at them)
        fused2 = vars(bb)["_hdn_fused"]
        ptrs = [b.data_ptr() for b in fused2.buffers()]
        other = Net().eval()
        for m in other.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.uniform_(-0.5, 0.5); m.running_var.uniform_(0.5, 2); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.5, 0.5)
        ref_other = other(x)
        bb.load_state_dict(other.state_dict())
        assert vars(bb)["_hdn_fused"] is fused2 and [b.data_ptr() for b in fused2.buffers()] == ptrs
        for a, b in zip(fused2(x), ref_other):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
        assert all(float((a - b).abs().max()) > 1e-3 for a, b in zip(fused2(x), ref))        # (really different weights)
        holder = nn.Module()                      # ... and through a parent's load_state_dict (ModelBuilder.load_state_dict in the reference's scripts)
        holder.backbone = bb
        holder.load_state_dict({"backbone." + k: v for k, v in Net().state_dict().items()})
        assert float((fused2(x)[0] - bb(x)[0]).abs().max()) <= 1e-5 * float(bb(x)[0].abs().max())
        BB.restore_similarity_model(model)
        assert len(bb._load_state_dict_post_hooks) == 0
        assert type(bb) is Net and "_hdn_fused" not in vars(bb) and type(model.neck) is Neck
        plain = types.SimpleNamespace(backbone=nn.Sequential(nn.Conv2d(3, 4, 3)), neck=None, neck_lp=None)
        assert BB.optimize_similarity_model(plain) == [] and type(plain.backbone) is nn.Sequential


def test_fold_conv_bn_arithmetic():
    """fold_conv_bn: eval-mode BatchNorm(conv(x)) == conv'(x) with the folded weights and shift (strided / dilated, with and without the
    convolution's own bias, affine-free BatchNorm), and the argument errors."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from hdn_amd.backbone import fold_conv_bn, FusedBottleneck
    torch.manual_seed(0)
    for bias, affine, kw in ((False, True, dict(stride=2, padding=0)), (True, True, dict(padding=2, dilation=2)), (False, False, dict(padding=1))):
        conv, bn = nn.Conv2d(5, 7, 3, bias=bias, **kw), nn.BatchNorm2d(7, affine=affine).eval()
        bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
        if affine:
            bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-1, 1)
        x = torch.randn(2, 5, 17, 17)
        w, b = fold_conv_bn(conv, bn)
        with torch.no_grad():
            ref = bn(conv(x))
            got = F.conv2d(x, w, b, conv.stride, conv.padding, conv.dilation)
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    with pytest.raises(ValueError):
        fold_conv_bn(nn.Conv2d(4, 4, 3, groups=2), nn.BatchNorm2d(4))
    with pytest.raises(ValueError):
        fold_conv_bn(nn.Conv2d(4, 4, 3), nn.BatchNorm2d(4, track_running_stats=False))
    with pytest.raises(ValueError):
        FusedBottleneck(nn.Conv2d(4, 4, 3))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_unchanged_launcher_runs_the_reference_test_script():
    """python -m hdn_amd.run /root/reference/tools/test.py ...: the reference's own benchmark script, byte for byte, with the
    hot path rebound before it starts; driven through ModelBuilder() and build_tracker(model) up to the dataset (the snapshot,
    POT-210 and cv2 do not exist in this container): build_tracker returns the device-resident tracker."""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    r = subprocess.run([sys.executable, "-m", "hdn_amd.run", "--no-build", "--preload", "launcher_preload:prepare",
                        "/root/reference/tools/test.py", "--dataset", "POT210", "--snapshot", "/nonexistent/model.pth",
                        "--config", "/root/reference/experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml"],
                       capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert "LAUNCHER_REACHED_LOAD_PRETRAIN True True True True" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "LAUNCHER_REACHED_DATASET True DeviceSimilarity True" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "hot-path sites rebound" in r.stderr
    # the script on disk is the reference's own
    assert r.returncode == 0


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_unchanged_launcher_runs_the_reference_demo_script():
    """python -m hdn_amd.run /root/reference/tools/demo.py ...: the reference's demo (tools/demo.py:95-106,168,172), byte for byte,
    up to its first OpenCV GUI call: the model it constructs carries the rebound modules and build_tracker(model) hands it the
    device-resident tracker (whose init / track_new it calls at :168 / :172 with the signatures hdn_amd.tracker keeps)."""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    r = subprocess.run([sys.executable, "-m", "hdn_amd.run", "--no-build", "--preload", "launcher_preload:prepare_demo",
                        "/root/reference/tools/demo.py", "--snapshot", "/nonexistent/model.pth", "--video", "/nonexistent/clip.mp4",
                        "--config", "/root/reference/experiments/tracker_homo_config/proj_e2e_GOT_unconstrained_v2.yaml"],
                       capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert "LAUNCHER_REACHED_LOAD_PRETRAIN True True True True" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DEMO_REACHED_GUI True DeviceSimilarity True True" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.returncode == 0
    # the calls demo.py makes on the tracker exist with its argument lists (:168 init(frame, bbox, rect, points, first_point), :172 track_new(idx, frame))
    import inspect
    from hdn_amd.tracker import DeviceTrackerHomo
    assert list(inspect.signature(DeviceTrackerHomo.init).parameters)[1:6] == ["img", "bbox", "poly", "gt_points", "first_point"]
    assert list(inspect.signature(DeviceTrackerHomo.track_new).parameters)[1:3] == ["fr_idx", "img"]


def test_launcher_argument_errors():
    from hdn_amd import run
    for argv in ([], ["--bogus", "x.py"], ["/nonexistent/script.py"], ["--reference-root"]):
        with pytest.raises(SystemExit) as e:
            run.main(argv)
        assert e.value.code == 2


@pytest.mark.parametrize("name", ["round1_bench_line.json", "round2_bench_line.json", "round3_bench_line.json", "round4_bench_line.json"])
def test_committed_bench_line_follows_the_contract(name):
    """profiles/roundN_bench_line.json is one output line of bench.py: the driver's fields, the roofline block of the
    dominant kernel and the CPU baseline must all be there and be self-consistent."""
    import json
    with open(os.path.join(ROOT, "profiles", name)) as f:
        d = json.load(f)
    if not name.startswith("round1"):
        fh = d["full_head"]
        assert fh["unit"] == "frames/s" and abs(fh["value"] - 64 * fh["n_gpus"] / (fh["ms_per_step"] * 1e-3)) <= 1e-6 * fh["value"]
        assert d["cpu_baseline"]["cores"] >= 1 and "1 thread pinned" in d["cpu_baseline"]["sample"]
        assert "xcorr_north_fft4_kernel" in d["roofline"]["kernel"]
    if name.startswith("round4"):
        sq = d["sequence"]      # BASELINE configs[3]: the end-to-end tracker loop with a production-shaped model
        assert sq["frames"] == 501 and sq["unit"] == "frames/s" and abs(sq["value"] - 1e3 / sq["ms_per_frame"]) <= 1e-6 * sq["value"]
        assert sq["host_syncs_per_frame"] == 1.0 and "DeviceTrackerHomo" in sq["tracker"] and "ResNet-50" in sq["model"]
        assert abs(sum(v for k, v in sq["share"].items() if k != "of_which_hip_correlations") - 1.0) <= 1e-6
        assert abs(sum(r["ms"] for r in sq["components"] if not r["stage"].startswith("  of which")) - sq["component_sum_ms"]) <= 1e-3   # (rows are rounded to 1e-4 ms)
        assert d["roofline"]["traffic"] is not None and "round4_pmc_hbm_traffic" in d["roofline"]["traffic_source"] or "round3_pmc_hbm_traffic" in d["roofline"]["traffic_source"]
    if name.startswith(("round3", "round4")):
        r3 = d["roofline"]
        assert r3["min_launch_ms"] <= r3["median_launch_ms"] and "in-step" in r3["timing"] and r3["timed_region_launch_ms"]["schedule"] == "after-north"
        assert abs(r3["timed_region_launch_ms"]["mean"] - r3["avg_launch_ms"]) <= 1e-9      # default schedule: the brackets ARE the timed region's
        assert "gpu_over_cpu" not in d and "median_of_5" in d["cpu_baseline"]["frames_per_s_1_thread"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 64 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0


def test_stride2_v2_weight_stream_layout():
    """pack_conv3x3s2_ds_v2: the fragment order include/hdn_hip.h documents for hdn_conv3x3s2_v2_f32, checked entry by entry."""
    from hdn_amd.trunk import pack_conv3x3s2_ds_v2
    g = torch.Generator().manual_seed(3)
    CI, CO = 64, 128
    w, wd = torch.randn(CO, CI, 3, 3, generator=g), torch.randn(CO, CI, 1, 1, generator=g)
    fr = pack_conv3x3s2_ds_v2(w, wd).view(torch.float16).reshape(CO // 64, CI // 32, 2, 10, 2, 2, 2, 32, 8)   # [nb, chunk, wk, step, nt, pc, g, n, e]
    assert fr.numel() == 2 * 10 * CI * CO
    for nb, ch, wk, st, nt, g_, n, e in ((0, 0, 0, 0, 0, 0, 0, 0), (1, 1, 1, 9, 1, 1, 31, 7), (0, 1, 0, 4, 1, 0, 5, 3), (1, 0, 1, 8, 0, 1, 17, 6)):
        co, ci = 64 * nb + 32 * nt + n, 32 * ch + 16 * wk + 8 * g_ + e
        wv = wd[co, ci, 0, 0] if st == 9 else w[co, ci, st // 3, st % 3]
        p0 = wv.to(torch.float16)
        assert fr[nb, ch, wk, st, nt, 0, g_, n, e] == p0 and fr[nb, ch, wk, st, nt, 1, g_, n, e] == ((wv - p0.float()) * 2048.0).to(torch.float16)
    with pytest.raises(ValueError):
        pack_conv3x3s2_ds_v2(w, wd[:, :32])


def test_fused_stem_host_side():
    """FusedStem (the trunk's first stage as one HIP kernel): weight layout, the library path for widths outside the kernel's
    range (works anywhere), and no CPU fallback inside the kernel's range."""
    from hdn_amd.trunk import FusedStem, fold_for_inference, resnet34_homo
    torch.manual_seed(2)
    f = fold_for_inference(resnet34_homo().eval(), channels_last=False)
    st = FusedStem(f.conv1, False)
    assert tuple(st.wT.shape) == (2, 7, 7, 64) and torch.equal(st.wT[1, 2, 3], f.conv1.weight[:, 1, 2, 3])
    x = torch.randn(1, 2, 6, 130)
    with torch.no_grad():
        ref = f.maxpool(f.relu(f.conv1(x)))
        assert torch.allclose(st(x), ref, atol=1e-6)
    with pytest.raises(Exception):
        st(torch.randn(1, 2, 9, 9))  # in range: needs the GPU library, never a CPU fallback
    g = fold_for_inference(resnet34_homo().eval(), channels_last=False, fused_stem=True)
    assert isinstance(g.conv1, FusedStem) and isinstance(g.maxpool, torch.nn.Identity)
    # the matrix-core form's weight stream (include/hdn_hip.h, hdn_trunk_stem_mfma_f32): [k step][n tile][piece][k half][n][8] fp16
    from hdn_amd.trunk import pack_stem_mfma
    w = f.conv1.weight.detach()
    fr = pack_stem_mfma(w).view(torch.float16).reshape(7, 2, 2, 2, 32, 8)
    assert st.wfrag.dtype == torch.int16 and st.wfrag.numel() == 7 * 2 * 2 * 64 * 8
    for s_, tile, g_, n, j in ((0, 0, 0, 0, 0), (3, 1, 1, 17, 6), (6, 1, 1, 31, 2), (3, 0, 0, 5, 3)):
        r = 2 * s_ + g_
        wv = w[32 * tile + n, r // 7, r % 7, j]
        p0 = wv.to(torch.float16)
        assert fr[s_, tile, 0, g_, n, j] == p0 and fr[s_, tile, 1, g_, n, j] == ((wv - p0.float()) * 2048.0).to(torch.float16)
    assert float(fr[:, :, :, :, :, 7].abs().max()) == 0.0                       # kx = 7: the zero tap that pads K to 112
    with pytest.raises(ValueError):
        pack_stem_mfma(torch.zeros(64, 3, 7, 7))


def test_committed_roofline_records_match_the_kernel_source():
    """The roofline block of bench.py quotes two committed records about the 31x31 (x) 61x61 kernel: its HBM traffic (rocprofv3 PMC passes,
    tools/pmc_traffic.py) and its instruction counts (tools/north_instr_count.py).  Both carry the SHA-256 of xcorr_fft.hip as measured; a
    later edit of that file makes bench.py print `traffic_source_matches_kernel_source: false` on the driver's line (round-5 VERDICT).
    Measurement experiments live in csrc/ablation/*.inc, outside the hashed file; this test holds the tree to the committed records."""
    import glob
    import hashlib
    import json
    have = hashlib.sha256(open(os.path.join(ROOT, "hdn_amd", "csrc", "xcorr_fft.hip"), "rb").read()).hexdigest()
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_hbm_traffic.json")))[-1]
    recs = [r for n, r in json.load(open(newest)).items() if "xcorr_north_fft4_kernel" in n and "traffic_calibrated_bytes" in r]
    assert recs, newest
    assert recs[0].get("kernel_source_sha256") == have, f"{os.path.basename(newest)} was measured on another xcorr_fft.hip: re-run tools/final_profile.sh"
    instr = json.load(open(os.path.join(ROOT, "profiles", "round6_north_instr.json")))
    assert instr["kernel_source_sha256"] == have, "re-run tools/north_instr_count.py"
    assert instr["valu_packed_per_pair"] <= instr["issued_per_pair"]
    # no measurement block inside a production translation unit
    for f in glob.glob(os.path.join(ROOT, "hdn_amd", "csrc", "*.hip")):
        assert "defined(HDN_ABLATION)" not in open(f).read(), f
    # the committed bench line of the round: the driver's contract fields, a roofline block that is consistent with itself and was produced on THIS kernel
    # source, the CPU baseline, and the round-6 additions (launch-carried brackets, measured copy rate, issue fractions, lock-step rows)
    d = json.loads([l for l in open(os.path.join(ROOT, "profiles", "round6_bench_line.json")) if l.startswith("{")][0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["algorithmic_bytes_per_launch"] == 5778432 * 64
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 and 0.3 < r["frac"] < 1.0
    assert r["traffic_source_matches_kernel_source"] is True and 0.98 <= r["traffic"] / r["algorithmic_bytes_per_launch"] <= 1.1
    assert r["instruction_count_matches_kernel_source"] is True and 0 < r["valu_frac"] < r["issue_frac"] < 1.0
    assert 1.0 <= r["shader_clock_GHz"] <= 2.6 and 3000 < r["measured_copy_GBps"] < 8000 and "hipExtLaunchKernelGGL" in r["bracket"]
    assert r["avg_launch_ms"] < d["ms_per_step"] and r["sustained_launch_ms"] >= r["min_launch_ms"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    rows = d["sequence"]["lockstep"]["rows"]
    assert rows and all(row["host_syncs_per_step"] == 1.0 and row["graph"] is True and row["frames_per_s"] > d["sequence"]["fps"] for row in rows)


def test_c_packers_match_the_layout_reference_bit_for_bit():
    """csrc/pack.hip (ABI 10: the weight streams are built by the library from plain [CO][CI][kh][kw] fp32 weights) against
    tests/reference_packers.py (the torch reshape / permute packers shipped until ABI 9) on random weights: every stream bit for bit, for every
    shape a kernel serves.  Host code only: runs without a GPU.  Also: sizes, range check (|w| >= 65,504 / NaN -> HDN_E_LIMIT -> ValueError),
    shapes no kernel serves."""
    import ctypes
    import reference_packers as R
    from hdn_amd import _lib, heads as HD, trunk as T
    g = torch.Generator().manual_seed(3)
    rn = lambda *s_: torch.randn(*s_, generator=g) * 0.1
    eq = lambda a, b: torch.equal(a.reshape(-1).cpu(), b.reshape(-1).cpu())
    for C in (64, 128, 256, 512):
        w = rn(C, C, 3, 3)
        assert eq(T.pack_conv3x3(w), R.pack_conv3x3(w)), C
        assert eq(T.pack_conv3x3_v2(w), R.pack_conv3x3_v2(w)), C
    for CI in (64, 128, 256):
        w, wd = rn(2 * CI, CI, 3, 3), rn(2 * CI, CI, 1, 1)
        assert eq(T.pack_conv3x3s2_ds(w, wd), R.pack_conv3x3s2_ds(w, wd)), CI
        assert eq(T.pack_conv3x3s2_ds_v2(w, wd), R.pack_conv3x3s2_ds_v2(w, wd)), CI
    ws = rn(64, 2, 7, 7)
    assert eq(T.pack_stem_mfma(ws), R.pack_stem_mfma(ws))
    for n, CO in ((3, 512), (1, 64), (4, 256)):
        wl = [rn(CO, 256, 3, 3) for _ in range(n)]
        assert eq(HD._pack_conv_search(wl), R._pack_conv_search(wl)), (n, CO)
    for G, H in ((6, 256), (2, 128), (8, 256)):
        w1 = rn(G, H, H)
        assert eq(HD._pack_w1(w1), R._pack_w1(w1)), (G, H)
    # values that exercise the split: exact halves, subnormal second pieces, the largest finite fp16, negative zero
    w = torch.zeros(64, 64, 3, 3)
    w.view(-1)[:6] = torch.tensor([65503.9, -65000.0, 1.0009765625, 6.1e-5, 5.9e-8, -0.0])
    assert eq(T.pack_conv3x3(w), R.pack_conv3x3(w))
    # range / shape errors
    for bad in (65504.0, float("nan"), -float("inf")):
        w = rn(64, 64, 3, 3)
        w[5, 6, 1, 2] = bad
        with pytest.raises(ValueError, match="fp16 range"):
            T.pack_conv3x3(w)
    with pytest.raises(ValueError):
        T.pack_conv3x3(rn(96, 96, 3, 3))
    with pytest.raises(ValueError):
        T.pack_conv3x3s2_ds(rn(128, 64, 3, 3), rn(128, 32, 1, 1))
    lib = _lib.load()
    assert lib.hdn_pack_conv3x3_bytes(256) == 2 * 2 * 256 * 256 * 9 and lib.hdn_pack_conv3x3_bytes(100) < 0
    assert lib.hdn_pack_conv3x3s2_ds_bytes(64) == 2 * 2 * 128 * 64 * 12 and lib.hdn_pack_stem_mfma_bytes() == 2 * 2 * 64 * 14 * 8
    buf = torch.empty(16, dtype=torch.int16)
    w = rn(64, 64, 3, 3).contiguous()
    assert lib.hdn_pack_conv3x3_f32(w.data_ptr(), 64, buf.data_ptr(), 32) != 0          # wrong size: refused, nothing written
    assert lib.hdn_pack_conv3x3_f32(None, 64, buf.data_ptr(), lib.hdn_pack_conv3x3_bytes(64)) != 0
