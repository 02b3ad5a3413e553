"""--preload hooks of tests/test_host_logic.py::test_unchanged_launcher_runs_the_reference_{test,demo}_script: make the reference's
tools/test.py / tools/demo.py importable in the build container (no cv2 / yacs / GPU there), lets it construct ModelBuilder() and call
build_tracker(model) (tools/test.py:65-72), and stops it at the next statement (the dataset, which does not exist here),
reporting what the model and the tracker were built from.  Not part of the product."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def prepare():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg

    mg.install_stubs()
    import types

    for name in ("shapely", "shapely.geometry", "tqdm", "glob2"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                m.Polygon = m.MultiPoint = m.tqdm = lambda *a, **k: None
                sys.modules[name] = m
    import hdn.utils.model_load as ml

    def stop_here(model, path):
        import hdn_amd
        from hdn_amd import install as hi
        import hdn.models.head.ban as ban
        print("LAUNCHER_REACHED_LOAD_PRETRAIN",
              isinstance(model.hm_net.ShareFeature, hdn_amd.PreShareFeature),
              isinstance(model.logpolar_instance, hdn_amd.STN_Polar),
              type(model).track_proj is hi._track_proj_method,
              ban.xcorr_depthwise is hdn_amd.xcorr_depthwise, flush=True)
        return model     # no snapshot in this container: carry on with the seeded weights

    ml.load_pretrain = stop_here
    import torch
    if not torch.cuda.is_available.__module__.startswith("torch"):   # make_golden's stub is active: no GPU in this container
        torch.cuda.current_device = lambda: 0                        # tools/test.py:74
    import toolkit.datasets as ds

    class _Factory:
        @staticmethod
        def create_dataset(**kw):
            import hdn_amd.tracker as T
            trk = sys._getframe(1).f_locals.get("tracker")     # main()'s local of tools/test.py:72
            print("LAUNCHER_REACHED_DATASET", type(trk) is T.DeviceTrackerHomo, type(trk.similarity).__name__,
                  trk.net is sys._getframe(1).f_locals["model"].hm_net, flush=True)
            raise SystemExit(0)

    ds.DatasetFactory = _Factory


def prepare_demo():
    """tools/demo.py (:95-106): cfg.merge_from_file, ModelBuilder(), load_pretrain(...).cuda().eval(), build_tracker(model), then its first
    OpenCV GUI call (cv2.namedWindow, :117) — where this hook reports what was built and stops (no display, no video, no cv2 here)."""
    prepare()
    import types
    cv2 = sys.modules["cv2"]
    cv2.WND_PROP_FULLSCREEN = 0

    def named_window(*a, **k):
        import hdn_amd.tracker as T
        loc = sys._getframe(1).f_locals                       # main()'s locals of tools/demo.py
        trk, model = loc.get("tracker"), loc.get("model")
        print("DEMO_REACHED_GUI", type(trk) is T.DeviceTrackerHomo, type(trk.similarity).__name__, trk.net is model.hm_net,
              trk.model is model, flush=True)
        raise SystemExit(0)

    cv2.namedWindow = named_window
