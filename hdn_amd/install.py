"""Rebind the reference's hot-path symbols to the HIP implementations.

    import hdn_amd.install as hi; hi.install()     # after the reference packages are importable

so that tools/test.py / tools/demo.py of the reference run unchanged.  Each rebinding site is the module
attribute the reference resolves at call time:

    hdn.core.xcorr.xcorr_depthwise{,_circular}           definitions
    hdn.models.head.ban.xcorr_depthwise                  ban.py:10 (from-import binding used at ban.py:76)
    hdn.models.head.ban_lp.xcorr_depthwise_circular      ban_lp.py:10 (used at ban_lp.py:38)
    ...Oneline_DLTv1.utils.{DLT_solve,transform,transformer}
    ...models.homo_model_builder.{DLT_solve,transform}   homo_model_builder.py:13
    hdn.models.model_builder_e2e_unconstrained_v2.{DLT_solve,Homo_STN}   :29-30
    ...preprocess.head['PreShareFeature']                registry used by get_pre(), preprocess/__init__.py:18-26
    ModelBuilder.track_proj                              replaced by the fused version
    hdn.models.logpolar.STN_Polar (+ its from-import in model_builder…:24)   device-resident log-polar sampler
    MultiBAN.forward / MultiCircBAN.forward              one correlation launch per head + cached template branch
    ModelBuilder.track_new_lp                            same ops, the zero `polar` argument cached on the device (:145: a pageable
                                                         host->device copy per frame, which a hipGraph capture cannot hold)
    ModelBuilder.template                                unchanged, then drops the heads' cached template-branch features
    hdn.tracker.tracker_builder.TRACKS['hdnTrackerHomoProje2e']   (install(tracker=True)) the device-resident tracker loop
"""
from __future__ import annotations

import importlib
import types

import torch

from . import heads, homo_model, homography, logpolar, share_feature, xcorr

_DLT = "homo_estimator.Deep_homography.Oneline_DLTv1"

REBINDINGS = (
    ("hdn.core.xcorr", "xcorr_depthwise", xcorr.xcorr_depthwise),
    ("hdn.core.xcorr", "xcorr_depthwise_circular", xcorr.xcorr_depthwise_circular),
    ("hdn.core.xcorr", "xcorr_fast", xcorr.xcorr_fast),
    ("hdn.core.xcorr", "xcorr_slow", xcorr.xcorr_slow),
    ("hdn.models.head.ban", "xcorr_fast", xcorr.xcorr_fast),
    ("hdn.models.head.ban", "xcorr_depthwise", xcorr.xcorr_depthwise),
    ("hdn.models.head.ban_lp", "xcorr_depthwise", xcorr.xcorr_depthwise),
    ("hdn.models.head.ban_lp", "xcorr_depthwise_circular", xcorr.xcorr_depthwise_circular),
    (_DLT + ".utils", "DLT_solve", homography.DLT_solve),
    (_DLT + ".utils", "transform", homography.transform),
    (_DLT + ".utils", "transformer", homography.transformer),
    (_DLT + ".models.homo_model_builder", "DLT_solve", homography.DLT_solve),
    (_DLT + ".models.homo_model_builder", "transform", homography.transform),
    ("hdn.models.model_builder_e2e_unconstrained_v2", "DLT_solve", homography.DLT_solve),
    ("hdn.models.model_builder_e2e_unconstrained_v2", "Homo_STN", homography.transform),
    # class rebinding: ModelBuilder.__init__ instantiates it (model_builder…:42), so install() must run first
    ("hdn.models.logpolar", "STN_Polar", logpolar.STN_Polar),
    ("hdn.models.model_builder_e2e_unconstrained_v2", "STN_Polar", logpolar.STN_Polar),
)


def _track_proj_method(self, data, tmp_mask):
    """ModelBuilder.track_proj with the fused stages; self.hm_net supplies ShareFeature/backbone/avgpool/fc."""
    return homo_model.track_proj(self.hm_net, data, tmp_mask)


def _track_new_lp_method(self, x, delta=[0, 0]):
    """ModelBuilder.track_new_lp (model_builder_e2e_unconstrained_v2.py:144-158), statement for statement, except that the
    all-zero `polar` lives on the device instead of being built on the host and uploaded every frame."""
    polar = getattr(self, "_hdn_polar0", None)
    if polar is None or polar.device != x.device or polar.shape[0] != x.shape[0]:
        polar = torch.zeros((x.shape[0], 2), dtype=torch.float32, device=x.device)
        object.__setattr__(self, "_hdn_polar0", polar)
    x_lp, grid = self.logpolar_instance(x, polar, delta)
    xf_lp = self.feature_extractor(x_lp)
    if hasattr(self, "neck_lp"):          # cfg.ADJUST.ADJUST (the constructor creates the necks under the same condition, :45-49)
        xf_lp = self.neck_lp(xf_lp)
    cls_lp, loc_lp = self.head_lp(self.zf_lp, xf_lp)
    return {"x_lp": x_lp, "cls_lp": cls_lp, "loc_lp": loc_lp, "grid": grid}


def _make_template_method(orig):
    def template(self, z):
        out = orig(self, z)
        for name in ("head", "head_lp"):
            if hasattr(self, name):
                heads.invalidate_template_cache(getattr(self, name))
        return out
    template._hdn_wraps = orig
    return template


_MISSING = object()
_saved = []  # (owner, attribute or dict key, original value, is_dict_item) in application order; undone by uninstall()


def _rebind(owner, attr, value, item=False):
    if item:
        _saved.append((owner, attr, owner.get(attr, _MISSING), True))
        owner[attr] = value
    else:
        _saved.append((owner, attr, owner.__dict__.get(attr, _MISSING) if isinstance(owner, type) else getattr(owner, attr, _MISSING), False))
        setattr(owner, attr, value)


def uninstall() -> int:
    """Undo every rebinding install() made (most recent first): the reference runs on its own PyTorch ops again, e.g.
    for training, which the HIP drop-ins (inference only, no autograd) do not support.  Returns the number undone."""
    n = 0
    while _saved:
        owner, attr, orig, item = _saved.pop()
        if item:
            if orig is _MISSING:
                owner.pop(attr, None)
            else:
                owner[attr] = orig
        elif orig is _MISSING:
            try:
                delattr(owner, attr)
            except AttributeError:
                pass
        else:
            setattr(owner, attr, orig)
        n += 1
    return n


def install(strict: bool = False, modules: dict = None, tracker: bool = False) -> list:
    """Apply the rebindings.  tracker=True also registers hdn_amd.tracker.DeviceTrackerHomo under
    TRACKS['hdnTrackerHomoProje2e'] (hdn/tracker/tracker_builder.py:12-19), so build_tracker(model) of tools/test.py:72 /
    tools/demo.py returns the device-resident loop (frames uploaded once, crops / warps / decodes as kernels, one host read per
    frame) instead of the host-side one.  `modules` (name -> module) lets tests supply stand-in modules; by default the
    real reference modules are imported.  Returns the list of (module, attribute) pairs that were rebound;
    with strict=True a site that cannot be imported raises instead of being skipped.  uninstall() reverses it."""
    done = []

    def get(name):
        if modules is not None:
            return modules.get(name)
        try:
            return importlib.import_module(name)
        except Exception:
            if strict:
                raise
            return None

    for mod_name, attr, fn in REBINDINGS:
        m = get(mod_name)
        if m is None:
            continue
        if not hasattr(m, attr) and strict:
            raise AttributeError(f"{mod_name}.{attr} not found: reference layout changed?")
        _rebind(m, attr, fn)
        done.append((mod_name, attr))

    pre = get(_DLT + ".preprocess")
    if pre is not None and isinstance(getattr(pre, "head", None), dict):
        _rebind(pre.head, "PreShareFeature", share_feature.PreShareFeature, item=True)
        done.append((_DLT + ".preprocess", "head['PreShareFeature']"))

    # heads: keep the reference's classes (and their weights), swap the schedule of forward()
    for mod_name, cls_name, circ in (("hdn.models.head.ban", "MultiBAN", False), ("hdn.models.head.ban_lp", "MultiCircBAN", True)):
        m = get(mod_name)
        if m is not None and hasattr(m, cls_name):
            klass = getattr(m, cls_name)
            if "_hdn_orig_forward" not in klass.__dict__ and "forward" in klass.__dict__:
                _rebind(klass, "_hdn_orig_forward", klass.__dict__["forward"])  # training-mode calls are deferred to it
            _rebind(klass, "forward", (lambda c: lambda self, z_fs, x_fs: heads.fused_forward(self, z_fs, x_fs, c))(circ))
            done.append((mod_name, cls_name + ".forward"))

    mb = get("hdn.models.model_builder_e2e_unconstrained_v2")
    if mb is not None and hasattr(mb, "ModelBuilder"):
        _rebind(mb.ModelBuilder, "track_proj", _track_proj_method)
        done.append(("hdn.models.model_builder_e2e_unconstrained_v2", "ModelBuilder.track_proj"))
        if "track_new_lp" in mb.ModelBuilder.__dict__:
            _rebind(mb.ModelBuilder, "track_new_lp", _track_new_lp_method)
            done.append(("hdn.models.model_builder_e2e_unconstrained_v2", "ModelBuilder.track_new_lp"))
        orig_t = mb.ModelBuilder.__dict__.get("template")
        if orig_t is not None and not hasattr(orig_t, "_hdn_wraps"):
            _rebind(mb.ModelBuilder, "template", _make_template_method(orig_t))
            done.append(("hdn.models.model_builder_e2e_unconstrained_v2", "ModelBuilder.template"))

    if tracker:
        tb = get("hdn.tracker.tracker_builder")
        if tb is None or not isinstance(getattr(tb, "TRACKS", None), dict):
            if strict:
                raise AttributeError("hdn.tracker.tracker_builder.TRACKS not found: reference layout changed?")
        else:
            from .tracker import DeviceTrackerHomo
            _rebind(tb.TRACKS, "hdnTrackerHomoProje2e", DeviceTrackerHomo, item=True)
            done.append(("hdn.tracker.tracker_builder", "TRACKS['hdnTrackerHomoProje2e']"))
            from .simi_tracker import DeviceTrackerSimi
            _rebind(tb.TRACKS, "hdnTracker", DeviceTrackerSimi, item=True)
            done.append(("hdn.tracker.tracker_builder", "TRACKS['hdnTracker']"))
    return done
