"""TRACKS['hdnTracker'] on the device (hdn_amd.simi_tracker, -m gpu): the similarity-only tracker with its per-frame template refresh
(hdn/tracker/hdn_tracker.py:109-301) against (a) what the reference's OWN init / track_new / update_template computed when they were executed
around the real ModelBuilder (tests/golden/tracker_loop_simi.npz, networks replayed) and (b) the CPU restatement of the loop
(oracle/tracker_oracle.SimiTrackerOracle, itself held to that fixture exactly) with live stand-in networks, eagerly and as one hipGraph."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class _ReplayNet(torch.nn.Module):
    """The reference's ModelBuilder interface with the recorded head maps of tracker_loop_simi.npz; template() keeps the crop it was handed
    and sets (dummy) template features the way ModelBuilder.template does (model_builder_e2e_unconstrained_v2.py:87-96)."""

    def __init__(self, g, prefix, dev):
        super().__init__()
        from tracker_loop_replay import ReplayModel
        self.anchor = torch.nn.Parameter(torch.zeros(1))
        self.replay = ReplayModel(g, prefix, to=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        self.zf = self.zf_lp = None
        self.templates = 0

    def template(self, z):
        self.replay.template(z)
        self.templates += 1
        self.zf = [z[:, 0:3, :7, :7].contiguous() * 0 + self.templates for _ in range(3)]
        self.zf_lp = [z[:, 3:6, :15, :15].contiguous() * 0 + self.templates for _ in range(3)]

    def track_new(self, x, delta=[0, 0]):
        return self.replay.track_new(x)

    def track_new_lp(self, x, delta=[0, 0]):
        return self.replay.track_new_lp(x, delta)


def test_device_simi_tracker_vs_executed_reference(dev):
    """Networks replayed; crops, decodes, the recurrences (centre, size, rot, lp_shift, scale, v, lost_count), the result dictionary and the
    template refresh (rotation of the resident first frame, crop + log-polar channels) run on the device.  uint8 crops bit for bit
    (CRC-32 of what the reference's model.template / track_new / track_new_lp were handed); indices, gates, float32-accumulated rot /
    lp_shift exactly; float64 geometry to 1e-9 relative; the polygon (np.cos / np.sin of a float32 angle in the reference) to 1e-4 px."""
    from conftest import load_golden
    from tracker_loop_replay import crc, tracker_loop_sequence
    from hdn_amd.simi_tracker import SimiTracker
    from hdn_amd.similarity import TrackerConfig
    g = load_golden("tracker_loop_simi")
    frames, init = tracker_loop_sequence(g)
    P = "s__"
    model = _ReplayNet(g, P, dev).to(dev).eval()
    cfg = TrackerConfig(instance_size=int(g[P + "instance_size"]), window_influence=float(g["window_influence"]))
    trk = SimiTracker(model, cfg=cfg, scale_score_thresh=float(g["scale_score_thresh"]))
    trk.init(frames[0], g["seq__bbox"].tolist(), g["seq__poly"].tolist(), np.array([g["seq__first_point"].tolist()]))
    rp = model.replay
    assert trk.init_s_z == float(g[P + "init__init_s_z"]) and trk.poly_shift_l == int(g[P + "init__poly_shift_l"])
    np.testing.assert_allclose(trk.channel_average, g[P + "init__channel_average"], rtol=1e-13)
    np.testing.assert_array_equal(rp.seen["z_crop"].cpu().numpy().astype(np.uint8), g[P + "init__z_crop"])
    syncs0 = trk.host_syncs
    worst = {"poly_px": 0.0, "geom_rel": 0.0}
    for i in range(1, int(g[P + "n_track"]) + 1):
        k = f"{P}f{i}__"
        rp.frame = i
        res = trk.track_new(i, frames[i], None, None)
        assert set(res) == {"bbox", "bbox_aligned", "best_score", "rot", "polygon"} and res["polygon"].shape == (4, 2)
        # what the networks were handed: both search crops and the refreshed template crop (the rotated first frame is inside the latter)
        assert crc(rp.seen["x_crop"].cpu().numpy().astype(np.uint8)) == int(g[k + "x_crop_crc"]), f"frame {i}: x_crop"
        assert crc(rp.seen["x_crop_moved"].cpu().numpy().astype(np.uint8)) == int(g[k + "x_crop_moved_crc"]), f"frame {i}: x_crop_moved"
        assert crc(rp.seen["z_crop"].cpu().numpy().astype(np.uint8)) == int(g[k + "z_crop_crc"]), f"frame {i}: refreshed template crop"
        st = trk.state.view(-1).cpu().numpy()
        ts = trk.track_state()
        assert int(st[6]) == int(g[k + "best_idx"]) and int(st[4]) == int(g[k + "stop"]) and int(st[18]) == int(g[k + "best_idx_lp"]), i
        # float32-accumulated quantities: exact; their numpy dtype as the reference ended up with
        assert ts["rot"] == float(g[k + "rot"]) and ts["lp_shift"][1] == float(g[k + "lp_shift1"]), (i, ts["rot"], float(g[k + "rot"]))
        assert ts["rot_is_float32"] == (str(g[k + "rot_kind"]) == "f") and ts["lp_shift_is_float32"] == (str(g[k + "lp_shift1_kind"]) == "f")
        assert ts["lost_count"] == int(g[k + "lost_count"]) and ts["last_lost"] == bool(g[k + "last_lost"])
        assert ts["window_scale_factor"] == float(g[k + "window_scale_factor"])
        assert float(res["best_score"]) == pytest.approx(float(g[k + "best_score"]), abs=1e-6)
        assert float(res["rot"]) == float(g[k + "res_rot"])
        for name, got, want in (("center", st[0:2], g[k + "center"]), ("center_pos", ts["center_pos"], g[k + "center_pos"]), ("size", ts["size"], g[k + "size"]),
                                ("scale", ts["scale"], g[k + "scale"]), ("v", ts["v"], g[k + "v"]), ("bbox", res["bbox"], g[k + "bbox"])):
            d = float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)))) / max(1.0, float(np.max(np.abs(want))))
            worst["geom_rel"] = max(worst["geom_rel"], d)
            assert d <= 1e-9, f"frame {i}: {name} differs by {d} (relative)"
        dp = float(np.max(np.abs(res["polygon"] - g[k + "polygon"])))
        da = float(np.max(np.abs(np.asarray(res["bbox_aligned"]) - g[k + "bbox_aligned"])))
        worst["poly_px"] = max(worst["poly_px"], dp, da)
        assert dp <= 1e-4 and da <= 2e-4, (i, dp, da)
        # the next frame's search window follows the new size (:176-188)
        seq = trk.seq.view(-1).cpu().numpy()
        if i < int(g[P + "n_track"]):
            assert seq[2] == float(g[f"{P}f{i + 1}__s_z"]) and seq[3] == float(g[f"{P}f{i + 1}__s_x"])
    assert trk.host_syncs - syncs0 == 2 * int(g[P + "n_track"])        # (one read per frame by track_new + this test's own track_state)
    assert model.templates == 1 + int(g[P + "n_track"])                # init + one refresh per frame
    print("device hdnTracker loop vs the executed reference:", worst)


def _standin(dev, **kw):
    import standin_model as SM
    from test_gpu_parity import _seeded_net
    from hdn_amd.similarity import TrackerConfig
    twin = SM.StandInSiamese(_seeded_net(), **kw).eval()
    cpu = SM.StandInSiameseCPU(twin)
    return twin.to(dev), cpu, TrackerConfig(cls_out_channels=twin.cls_out)


def test_simi_tracker_stream_device_vs_cpu_loop_eager_and_graph(dev):
    """Live (stand-in) networks: the device loop — eager and as ONE hipGraph per frame — against the CPU restatement frame by frame.  The
    template refresh feeds back: the template features differ from frame to frame, and a tracker whose refresh is disabled drifts away."""
    from synth_sequence import make_sequence
    from hdn_amd.simi_tracker import SimiTracker
    from oracle.tracker_oracle import SimiTrackerOracle
    frames, corners, init = make_sequence(n_frames=11, frame_hw=(360, 640), target_wh=(150, 100), seed=7)
    twin, cpu, cfg = _standin(dev, loc_scale_lp=0.3)
    ref = SimiTrackerOracle(cpu)
    fp = np.array([init["first_point"]])
    ref.init(frames[0], init["bbox"], init["poly"], fp)
    eager = SimiTracker(twin, cfg=cfg)
    eager.init(frames[0], init["bbox"], init["poly"], init["gt_points"], fp)      # (the launchers' five-argument spelling)
    twin_g = copy.deepcopy(twin)
    graphed = SimiTracker(twin_g, cfg=cfg, graph=True)
    graphed.init(frames[0], init["bbox"], init["poly"], fp)
    z0 = [t.clone() for t in twin.zf]
    worst_e = worst_g = 0.0
    rots = []
    for i in range(1, len(frames)):
        a, gph, b = eager.track_new(i, frames[i]), graphed.track_new(i, frames[i]), ref.track_new(i, frames[i])
        de = float(np.max(np.abs(a["polygon"] - b["polygon"])))
        dg = float(np.max(np.abs(gph["polygon"] - b["polygon"])))
        worst_e, worst_g = max(worst_e, de), max(worst_g, dg)
        # same argmax cells and gates -> sub-pixel agreement; a cell off would be 8 px * s_z / 127
        assert de <= (2e-3 if i <= 3 else 5e-2) and dg <= (2e-3 if i <= 3 else 5e-2), (i, de, dg)
        assert abs(float(a["rot"]) - float(b["rot"])) <= 2e-4 and abs(float(a["best_score"]) - float(b["best_score"])) <= 1e-4
        np.testing.assert_allclose(a["bbox"], b["bbox"], atol=5e-2)
        rots.append(float(b["rot"]))
    assert graphed._graph is not None
    assert max(abs(r) for r in rots) > 1e-3, "the stand-in must rotate the template for this test to mean anything"
    assert not torch.equal(z0[0], twin.zf[0]), "the refreshed template features must differ from the first frame's"
    # the static buffers ARE the model's template (refreshed in place): same storage after ten refreshes
    assert twin.zf[0].data_ptr() == z0[0].data_ptr() or twin.zf[0].data_ptr() == eager._zf_static[0][0].data_ptr()
    print(f"hdnTracker device loop vs CPU loop, worst polygon distance: eager {worst_e:.2e} px, hipGraph {worst_g:.2e} px; final rot {rots[-1]:+.4f}")
    # sync=False: the device record, no host read
    s0 = eager.host_syncs
    rec = eager.track_new(99, frames[-1], sync=False)["record"]
    assert rec.is_cuda and rec.shape == (20,) and eager.host_syncs == s0


def test_device_tracker_simi_production_shape(dev):
    """DeviceTrackerSimi(model) — what install(tracker=True) registers under TRACKS['hdnTracker'] — around the production-shaped stand-in
    (ResNet-50 on PyTorch-ROCm, 256-channel heads: 4 backbone passes per frame incl. the template refresh), one hipGraph per frame, against the CPU loop."""
    import production_standin as PS
    from synth_sequence import make_sequence
    from test_gpu_parity import _seeded_net
    from hdn_amd.simi_tracker import DeviceTrackerSimi
    from oracle.tracker_oracle import SimiTrackerOracle
    frames, corners, init = make_sequence(n_frames=6, frame_hw=(480, 854), target_wh=(200, 140), seed=20260928)
    twin = PS.ProductionStandIn(_seeded_net())
    twin.calibrate(*PS.calibration_crops(frames, init))
    ref = SimiTrackerOracle(PS.ProductionStandInCPU(twin))
    fp = np.array([init["first_point"]])
    ref.init(frames[0], init["bbox"], init["poly"], fp)
    model = twin.to(dev).eval()
    trk = DeviceTrackerSimi(model)
    assert trk.use_graph is True and trk.folded == ["backbone", "neck", "neck_lp"]
    trk.init(frames[0], init["bbox"], init["poly"], fp)
    errs = []
    for i in range(1, len(frames)):
        a, b = trk.track_new(i, frames[i]), ref.track_new(i, frames[i])
        errs.append(float(np.max(np.abs(a["polygon"] - b["polygon"]))))
    print("production-shaped hdnTracker device loop vs CPU loop, polygon distance (px):", " ".join(f"{e:.1e}" for e in errs))
    assert trk._graph is not None
    assert errs[0] <= 5e-2 and max(errs) <= 0.5, errs        # (50 fp32 layers, MIOpen vs oneDNN; a decode cell off would be >= 10 px)
    # the template refresh ran as ONE backbone pass over (crop, log-polar crop): accepted for this model by the check at init, and the
    # tracker that calls the model's own template() — two passes — gives the same polygons
    assert trk.batch_template and trk._batched_ok
    two = DeviceTrackerSimi(model, batch_template=False)
    two.init(frames[0], init["bbox"], init["poly"], fp)
    assert not two._batched_ok
    one = DeviceTrackerSimi(model)
    one.init(frames[0], init["bbox"], init["poly"], fp)
    d = [float(np.max(np.abs(one.track_new(i, frames[i])["polygon"] - two.track_new(i, frames[i])["polygon"]))) for i in range(1, len(frames))]
    print("one-pass template refresh vs template(): polygon distance (px):", " ".join(f"{e:.1e}" for e in d))
    assert max(d) <= 2e-2, d


def test_batched_template_is_refused_for_models_it_does_not_describe(dev):
    """_check_batched_template: a model without feature_extractor / with one neck only keeps its own template(); a model whose template() is NOT
    `neck(backbone(z)), neck_lp(backbone(z_lp))` is caught by the numerical comparison on the first crop."""
    from hdn_amd.simi_tracker import SimiTracker

    class Net(torch.nn.Module):
        def __init__(self, twist):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 8, 3, bias=False)
            self.neck = torch.nn.Conv2d(8, 8, 1)
            self.neck_lp = torch.nn.Conv2d(8, 8, 1)
            self.twist = twist

        def feature_extractor(self, x):
            return self.conv(x)

        def template(self, z):
            self.zf = self.neck(self.feature_extractor(z[:, 0:3]))
            self.zf_lp = self.neck_lp(self.feature_extractor(z[:, 3:6])) * self.twist

    torch.manual_seed(5)
    z = torch.randn(2, 6, 31, 31, device=dev)
    for twist, want in ((1.0, True), (1.001, False)):
        net = Net(twist).to(dev).eval()
        t = SimiTracker(net, batch_template=True)
        t._template(z, first=True)
        assert t._batched_ok is want, (twist, t._batched_ok)
        ref = Net(twist).to(dev).eval()
        ref.load_state_dict(net.state_dict())
        ref.template(z)
        torch.testing.assert_close(net.zf, ref.zf, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(net.zf_lp, ref.zf_lp, rtol=1e-5, atol=1e-6)
    net = Net(1.0).to(dev).eval()
    del net.neck_lp
    net.template = lambda z: (setattr(net, "zf", net.conv(z[:, 0:3])), setattr(net, "zf_lp", net.conv(z[:, 3:6])))
    t = SimiTracker(net, batch_template=True)
    t._template(z, first=True)
    assert t._batched_ok is False
    t = SimiTracker(Net(1.0).to(dev).eval(), batch_template=False)
    t._template(z, first=True)
    assert t._batched_ok is False


class _ScriptedNet(torch.nn.Module):
    """ModelBuilder's interface with SCRIPTED head maps: frame i's maps put the translation peak in a chosen cell with a chosen offset and make the
    log-polar head answer a chosen (scale, rotation) — so that a test can steer the recurrences of hdnTracker.track_new into their rare branches.
    Works on CPU tensors (the oracle's loop) and on the device alike; keeps the crops it was handed."""

    def __init__(self, script, dev=None):
        super().__init__()
        self.anchor = torch.nn.Parameter(torch.zeros(1))
        self.script, self.dev, self.frame, self.seen = script, dev, 0, {}
        self.zf = self.zf_lp = None

    def _t(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a, np.float32))
        return t.to(self.dev) if self.dev is not None else t

    def template(self, z):
        self.seen["z_crop"] = z
        self.zf = [z[:, 0:3, :7, :7].contiguous() * 0 for _ in range(3)]
        self.zf_lp = [z[:, 3:6, :15, :15].contiguous() * 0 for _ in range(3)]

    def track_new(self, x, delta=[0, 0]):
        self.seen["x_crop"] = x
        s = self.script[self.frame]
        cls = np.zeros((1, 2, 25, 25), np.float32)
        cls[0, 1] = -6.0
        cls[0, 1].reshape(-1)[s["cell"]] = s.get("logit", 6.0)
        loc = np.zeros((1, 2, 25, 25), np.float32)
        loc[0, 0], loc[0, 1] = s["dx"], s["dy"]
        return {"cls": self._t(cls), "loc_c": self._t(loc)}

    def track_new_lp(self, x, delta=[0, 0]):
        self.seen["x_crop_moved"] = x
        s = self.script[self.frame]
        cls = np.zeros((1, 2, 13, 13), np.float32)
        cls[0, 1] = -6.0
        cls[0, 1].reshape(-1)[s.get("cell_lp", 84)] = s.get("logit_lp", 6.0)
        loc = np.zeros((1, 4, 13, 13), np.float32)
        loc[0, 0], loc[0, 2] = s["lp_scale"], s["lp_rot"]     # sim_lp[0] = exp((px - 8 lp_scale) mag), sim_lp[2] = (py - 8 lp_rot) 2 pi / 127
        return {"cls_lp": self._t(cls), "loc_lp": self._t(loc)}


def test_simi_tracker_rare_branches_vs_cpu_loop(dev):
    """The branches a well-behaved sequence never takes, forced by scripted head maps and held to the CPU restatement of hdnTracker.track_new
    (itself pinned to the executed reference): rotation past +2 pi and below -2 pi (both wraps, hdn_tracker.py:264-269, in float32 as numpy has it),
    the size clamps at the image (`min(width, W)`), at 10 px and at 10 * w0 / h0 (:249-250), a log-polar-gated frame, frames under
    SCALE_SCORE_THRESH (the lost_count bookkeeping, :214-222) — and the crops the kernels cut at the window sizes those produce (search windows of
    a few pixels and of several frame widths, all padding)."""
    from synth_sequence import make_sequence
    from tracker_loop_replay import crc
    from hdn_amd.simi_tracker import SimiTracker
    from hdn_amd.similarity import TrackerConfig
    from oracle.tracker_oracle import SimiTrackerOracle
    frames, corners, init = make_sequence(n_frames=3, frame_hw=(240, 320), target_wh=(80, 50), seed=11)
    g = np.random.default_rng(4)
    base = dict(cell=312, dx=0.3, dy=-0.2, lp_scale=0.0, lp_rot=0.0)
    script = {}
    n = 0
    def add(k, **kw):
        nonlocal n
        for _ in range(k):
            n += 1
            script[n] = dict(base, **kw)
    add(4, lp_rot=-5.0)                  # +1.98 rad per frame (py = 0 at the centre row): past +2 pi on the 4th -> wrap down
    add(8, lp_rot=5.0)                   # -1.98 rad per frame: below -2 pi -> wrap up
    add(3, lp_scale=-3.0)                # sim_lp[0] = exp(240 mag) = 2.2 per frame: width -> clamped at the frame's 320, height at 240
    add(1, logit_lp=-9.0)                # log-polar gate: sim_lp = [1, 1, 0, 0]
    add(6, lp_scale=4.0)                 # 0.35 per frame: height -> 10, width -> 10 * w0 / h0
    add(2, logit=-3.0)                   # pscore under SCALE_SCORE_THRESH: lost_count
    add(2, cell=0, dx=-3.0, dy=2.5)      # the corner cell: a long jump of the window (padding on two sides)
    cpu_model, dev_model = _ScriptedNet(script), _ScriptedNet(script, dev).to(dev)
    ref = SimiTrackerOracle(cpu_model)
    trk = SimiTracker(dev_model, cfg=TrackerConfig())
    fp = np.array([init["first_point"]])
    poly = list(init["poly"][:4]) + [0.3]          # a rotated initial box: python float theta
    ref.init(frames[0], init["bbox"], poly, fp)
    trk.init(frames[0], init["bbox"], poly, fp)
    seen = {"wrap_down": 0, "wrap_up": 0, "w_max": 0, "h_min": 0, "w_min": 0, "gated": 0, "lost": 0}
    lo = 10 * init["poly"][2] / init["poly"][3]
    prev_rot = 0.3
    for i in range(1, n + 1):
        cpu_model.frame = dev_model.frame = i
        img = frames[i % len(frames)]
        b = ref.track_new(i, img)
        a = trk.track_new(i, img)
        ts = trk.track_state()
        for name, got, want in (("center_pos", ts["center_pos"], ref.center_pos), ("size", ts["size"], ref.size), ("scale", ts["scale"], float(ref.scale)),
                                ("v", ts["v"], float(ref.v)), ("bbox", a["bbox"], b["bbox"])):
            d = float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)))) / max(1.0, float(np.max(np.abs(np.asarray(want, np.float64)))))
            assert d <= 1e-9, (i, name, got, want)
        assert ts["rot"] == float(ref.rot) and ts["lp_shift"][1] == float(ref.lp_shift[1]), (i, ts["rot"], float(ref.rot))
        assert ts["rot_is_float32"] == isinstance(ref.rot, np.float32) and ts["lp_shift_is_float32"] == isinstance(ref.lp_shift[1], np.float32)
        assert ts["lost_count"] == int(ref.lost_count) and ts["last_lost"] == bool(ref.last_lost) and ts["window_scale_factor"] == float(ref.window_scale_factor)
        assert float(np.max(np.abs(a["polygon"] - b["polygon"]))) <= 2e-3 and float(np.max(np.abs(np.asarray(a["bbox_aligned"]) - np.asarray(b["bbox_aligned"])))) <= 4e-3, i
        # the crops the networks were handed, at whatever window size the recurrences produced
        for k in ("x_crop", "x_crop_moved", "z_crop"):
            assert crc(dev_model.seen[k].cpu().numpy().astype(np.uint8)) == crc(cpu_model.seen[k].numpy().astype(np.uint8)), (i, k, float(ref.trace["s_x"]))
        r = float(ref.rot)
        seen["wrap_down"] += int(r < prev_rot - 3.0)
        seen["wrap_up"] += int(r > prev_rot + 3.0)
        prev_rot = r
        seen["w_max"] += int(ref.size[0] == 320 and ref.size[1] == 240)
        seen["h_min"] += int(ref.size[1] == 10)
        seen["w_min"] += int(abs(ref.size[0] - lo) < 1e-12)
        seen["gated"] += int(list(ref.trace["sim_lp"]) == [1, 1, 0, 0])
        seen["lost"] += int(float(ref.window_scale_factor) == 1.5)
    assert all(v >= 1 for v in seen.values()), seen
    print("rare branches taken:", seen)


def test_batched_simi_tracker_equals_single_runs_and_cpu_loop(dev):
    """BatchedSimiTracker (n = 3 sequences of different targets in lock step, eager and as one hipGraph per step): every sequence against its own SimiTracker
    run (same kernels at B = 1) and against the CPU restatement of hdnTracker's loop; one host read per step for all of them."""
    from synth_sequence import make_sequence
    from hdn_amd.simi_tracker import BatchedSimiTracker, SimiTracker
    from oracle.tracker_oracle import SimiTrackerOracle
    n, T = 3, 8
    sizes = [(150, 100), (120, 90), (170, 110)]
    seqs = [make_sequence(n_frames=T, frame_hw=(360, 640), target_wh=sizes[b], seed=80 + b) for b in range(n)]
    twin, cpu, cfg = _standin(dev, loc_scale_lp=0.3)
    single, ref = [], []
    for frames, _, init in seqs:
        fp = np.array([init["first_point"]])
        t, r = SimiTracker(twin, cfg=cfg), SimiTrackerOracle(cpu)
        t.init(frames[0], init["bbox"], init["poly"], fp)
        r.init(frames[0], init["bbox"], init["poly"], fp)
        single.append([t.track_new(i, frames[i]) for i in range(1, T)])
        ref.append([r.track_new(i, frames[i]) for i in range(1, T)])
    args = ([s[0][0] for s in seqs], [s[2]["bbox"] for s in seqs], [s[2]["poly"] for s in seqs], [np.array([s[2]["first_point"]]) for s in seqs])
    worst = {}
    for graph in (False, True):
        bt = BatchedSimiTracker(twin, n, cfg=cfg, graph=graph)
        bt.init(*args)
        assert [int(z.shape[0]) for z in twin.zf] == [n] * len(twin.zf)
        s0 = bt.host_syncs
        ws = wc = 0.0
        for i in range(1, T):
            res = bt.track_new(i, [s[0][i] for s in seqs])
            assert len(res) == n and set(res[0]) == {"bbox", "bbox_aligned", "best_score", "rot", "polygon"}
            for b in range(n):
                ds = float(np.max(np.abs(res[b]["polygon"] - single[b][i - 1]["polygon"])))
                dc = float(np.max(np.abs(res[b]["polygon"] - ref[b][i - 1]["polygon"])))
                ws, wc = max(ws, ds), max(wc, dc)
                assert ds <= (2e-3 if i <= 3 else 5e-2) and dc <= (2e-3 if i <= 3 else 5e-2), (graph, i, b, ds, dc)
                assert abs(float(res[b]["rot"]) - float(single[b][i - 1]["rot"])) <= 2e-4
        assert bt.host_syncs - s0 == T - 1 and (bt._graph is not None) == graph
        worst[graph] = (ws, wc)
        st = bt.track_state()
        assert len(st) == n and all(s_["frames"] == T - 1 for s_ in st)
    print("batched hdnTracker loop (n = 3), worst polygon distance (to the B = 1 runs, to the CPU loop):", {("hipGraph" if k else "eager"): tuple(f"{v:.1e}" for v in w) for k, w in worst.items()})
    with pytest.raises(ValueError):
        bt.track_new(99, [seqs[0][0][1]])
