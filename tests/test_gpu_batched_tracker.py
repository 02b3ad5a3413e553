"""N sequences in lock step on one GPU (hdn_amd.batched_tracker) — the reference's only inference-time parallelism is several
videos at once (tools/test.py:91-103: hand-split video ranges, one process each).  Every sequence of a batch is held to (a) its own
B = 1 run through hdn_amd.tracker.HomoTracker (bit-exact frame kernels; the networks are the same code at another batch size) and
(b) the CPU restatement of the reference's loop (oracle/tracker_oracle.py, hdn_tracker_proj_e2e.py:141-285)."""
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


def _sequences(n, n_frames, frame_hw=(360, 640)):
    """n different sequences of one frame size: different textures, walks AND target sizes (so every per-sequence record differs)."""
    from synth_sequence import make_sequence
    sizes = [(150, 100), (120, 90), (170, 110), (140, 140), (100, 130)]
    return [make_sequence(n_frames=n_frames, frame_hw=frame_hw, target_wh=sizes[b % len(sizes)], seed=70 + b) for b in range(n)]


def _init_args(seqs):
    return ([s[0][0] for s in seqs], [s[2]["bbox"] for s in seqs], [s[2]["poly"] for s in seqs], [s[2]["gt_points"] for s in seqs],
            [s[2]["first_point"] for s in seqs])


def test_batched_frame_kernels_bit_exact_per_sequence(dev):
    """hdn_frame_warp_perspective / warp_affine_cubic / subwindow at [N,H,W,3] with per-sequence parameter records: sequence b of the
    batch == the same call on frame b alone, bit for bit (blockIdx.y only selects the data)."""
    from hdn_amd import frame as FR
    g = np.random.default_rng(5)
    n, H, W = 4, 180, 320
    frames = torch.from_numpy(g.integers(0, 256, (n, H, W, 3), dtype=np.uint8)).to(dev)
    Hs = np.tile(np.eye(3), (n, 1, 1))
    Hs[:, :2, 2] = g.normal(0, 6, (n, 2))
    Hs[:, :2, :2] += g.normal(0, 0.03, (n, 2, 2))
    Hs[:, 2, :2] = g.normal(0, 1e-5, (n, 2))
    Hd = torch.from_numpy(Hs.reshape(n, 9)).to(dev)
    wb = FR.warp_perspective(frames, Hd)
    assert wb.shape == frames.shape and wb.dtype == torch.uint8
    for b in range(n):
        assert torch.equal(wb[b], FR.warp_perspective(frames[b], Hd[b])), b
    rot = np.zeros((n, 6))
    for b in range(n):
        a = g.normal(0, 0.1)
        rot[b] = [np.cos(a), -np.sin(a), g.normal(0, 3), np.sin(a), np.cos(a), g.normal(0, 3)]
    Rd = torch.from_numpy(rot).to(dev)
    rb = FR.warp_affine_cubic(frames, Rd)
    for b in range(n):
        assert torch.equal(rb[b], FR.warp_affine_cubic(frames[b], Rd[b])), b
    params = np.concatenate([g.uniform(60, 120, (n, 2)), g.uniform(80, 200, (n, 1)), g.uniform(90, 130, (n, 3))], axis=1)
    Pd = torch.from_numpy(params).to(dev)
    for sz, islog in ((127, 0), (255, 0), (127, 1)):
        cb = FR.get_subwindow(frames, None, sz, None, None, params=Pd, islog=islog)
        assert cb.shape[0] == n
        for b in range(n):
            one = FR.get_subwindow(frames[b], None, sz, None, None, params=Pd[b], islog=islog)
            assert torch.equal(cb[b], one[0]), (sz, islog, b)
    sb = FR.get_search_info(frames, None, None, None, model_sz=127, params=Pd)
    for b in range(n):
        assert torch.equal(sb[b], FR.get_search_info(frames[b], None, None, None, model_sz=127, params=Pd[b])[0]), b


def test_batched_homography_tracker_equals_single_runs_and_cpu_loop(dev):
    """BatchedHomoTracker without a similarity branch, n = 3: every sequence against its own HomoTracker run and the CPU loop."""
    from synth_sequence import success_4pts_error
    from test_gpu_parity import _seeded_net
    from hdn_amd.batched_tracker import BatchedHomoTracker
    from hdn_amd.tracker import HomoTracker
    from oracle.tracker_oracle import HomoTrackerOracle
    n, T = 3, 8
    seqs = _sequences(n, T)
    net = _seeded_net()
    net_cpu = copy.deepcopy(net)
    net = net.to(dev)
    sd = {k: v.clone() for k, v in net_cpu.ShareFeature.state_dict().items()}
    single, cpu = [], []
    for frames, _, init in seqs:
        t = HomoTracker(net)
        r = HomoTrackerOracle(sd, lambda f: net_cpu.fc(net_cpu.avgpool(net_cpu.backbone(f)).flatten(1)))
        for x in (t, r):
            x.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
        single.append([t.track_new(i, frames[i]) for i in range(1, T)])
        cpu.append([r.track_new(i, frames[i]) for i in range(1, T)])
    bt = BatchedHomoTracker(net, n)
    bt.init(*_init_args(seqs))
    s0 = bt.host_syncs
    worst_single, worst_cpu = 0.0, 0.0
    for i in range(1, T):
        res = bt.track_new(i, [s[0][i] for s in seqs])
        assert len(res) == n
        for b in range(n):
            assert set(res[b]) == {"bbox_aligned", "best_score", "polygon", "points", "bbox"} and res[b]["points"].shape == (4, 2)
            es = success_4pts_error(res[b]["points"], single[b][i - 1]["points"])
            ec = success_4pts_error(res[b]["points"], cpu[b][i - 1]["points"])
            worst_single, worst_cpu = max(worst_single, es), max(worst_cpu, ec)
            # the trunk at B = 3 is the large-batch-capable form of the same kernels as at B = 1 (chained form there): rounding only
            assert es <= (2e-4 if i <= 3 else 5e-3), (i, b, es)
            assert ec <= (5e-4 if i <= 3 else 5e-3), (i, b, ec)
    print(f"batched (n={n}) homography loop: worst corner distance to the B=1 runs {worst_single:.2e} px, to the CPU loop {worst_cpu:.2e} px")
    assert bt.host_syncs - s0 == T - 1           # ONE host read per step for all n sequences
    with pytest.raises(ValueError):
        bt.track_new(99, [seqs[0][0][1]])        # n frames per step, always
    with pytest.raises(ValueError):
        bt.track_new(99, [seqs[0][0][1]] * (n - 1) + [seqs[0][0][1][:100]])     # one frame size per step


def _similarity_twin(dev, **standin_kw):
    import standin_model as SM
    from test_gpu_parity import _seeded_net
    from hdn_amd.similarity import TrackerConfig
    net = _seeded_net()
    twin = SM.StandInSiamese(net, **standin_kw).eval()
    cpu = SM.StandInSiameseCPU(twin)
    net_cpu = copy.deepcopy(net)
    return twin.to(dev), cpu, net_cpu, TrackerConfig(cls_out_channels=twin.cls_out)


def test_batched_tracker_with_similarity_equals_single_runs_and_cpu_loop(dev):
    """The whole frame body (stabilising warp, two crops + heads + decodes, rotate-back, homography crop, track_proj, accumulation) at
    n = 4, eagerly and as ONE hipGraph per step: per sequence the B = 1 device loop's corners and the CPU loop's."""
    from synth_sequence import success_4pts_error
    from hdn_amd.batched_tracker import BatchedHomoTracker
    from hdn_amd.similarity import DeviceSimilarity
    from hdn_amd.tracker import HomoTracker
    from oracle.tracker_oracle import HomoTrackerOracle, SimilarityOracle
    n, T = 4, 9
    seqs = _sequences(n, T)
    twin, cpu_model, net_cpu, cfg = _similarity_twin(dev)
    sd = {k: v.clone() for k, v in net_cpu.ShareFeature.state_dict().items()}
    single, cpu, single_state = [], [], []
    for frames, _, init in seqs:
        t = HomoTracker(twin.hm_net, similarity=DeviceSimilarity(twin, cfg), cfg=cfg)
        r = HomoTrackerOracle(sd, lambda f: net_cpu.fc(net_cpu.avgpool(net_cpu.backbone(f)).flatten(1)), similarity=SimilarityOracle(cpu_model))
        for x in (t, r):
            x.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
        rs, st = [], []
        for i in range(1, T):
            rs.append(t.track_new(i, frames[i]))
            st.append(t.similarity.state.view(-1).cpu().numpy().copy())
        single.append(rs)
        single_state.append(st)
        cpu.append([r.track_new(i, frames[i]) for i in range(1, T)])
    eager = BatchedHomoTracker(twin.hm_net, n, similarity=DeviceSimilarity(twin, cfg), cfg=cfg)
    eager.init(*_init_args(seqs))
    assert [int(z.shape[0]) for z in twin.zf] == [n] * len(twin.zf)          # the model holds n templates
    res_e = [eager.track_new(i, [s[0][i] for s in seqs]) for i in range(1, T)]
    states = eager.similarity.state.cpu().numpy()
    graphed = BatchedHomoTracker(twin.hm_net, n, similarity=DeviceSimilarity(twin, cfg), cfg=cfg, graph=True)
    graphed.init(*_init_args(seqs))
    res_g = [graphed.track_new(i, [s[0][i] for s in seqs]) for i in range(1, T)]
    assert graphed._graph is not None
    we = wg = wc = 0.0
    for i in range(T - 1):
        for b in range(n):
            es = success_4pts_error(res_e[i][b]["points"], single[b][i]["points"])
            eg = success_4pts_error(res_g[i][b]["points"], single[b][i]["points"])
            ec = success_4pts_error(res_e[i][b]["points"], cpu[b][i]["points"])
            we, wg, wc = max(we, es), max(wg, eg), max(wc, ec)
            assert es <= (2e-4 if i < 3 else 5e-3), (i, b, es)
            assert eg <= (1e-3 if i < 3 else 2e-2), (i, b, eg)
            assert ec <= (5e-4 if i < 3 else 5e-3), (i, b, ec)
            assert abs(float(res_e[i][b]["best_score"]) - float(single[b][i]["best_score"])) <= 1e-5
    # the decoded similarity of the last step, per sequence: same argmax cells, gates, centre
    for b in range(n):
        a, s = states[b], single_state[b][-1]
        assert a[6] == s[6] and a[18] == s[18] and a[4] == s[4], (b, a[:20], s[:20])
        np.testing.assert_allclose(a[2:4], s[2:4], atol=2e-2)
    print(f"batched (n={n}) loop with similarity: worst corner distance to the B=1 runs {we:.2e} px (eager) {wg:.2e} px (hipGraph), to the CPU loop {wc:.2e} px")


def test_batched_tracker_per_sequence_score_gate(dev):
    """The reference's gate (hdn_tracker_proj_e2e.py:251-258 via the `[0][0]` scores, model_builder_e2e_unconstrained_v2.py:213-216) is
    taken per sequence: one sequence fed a frame with no target (gate closes, H_total keeps its value) does not affect its neighbours."""
    from hdn_amd.batched_tracker import BatchedHomoTracker
    from hdn_amd.tracker import HomoTracker
    from test_gpu_parity import _seeded_net
    n, T = 3, 4
    seqs = _sequences(n, T)
    net = _seeded_net().to(dev)
    bt = BatchedHomoTracker(net, n)
    bt.init(*_init_args(seqs))
    ref = BatchedHomoTracker(net, n)
    ref.init(*_init_args(seqs))
    noise = np.random.default_rng(3).integers(0, 256, seqs[0][0][0].shape, dtype=np.uint8)
    for i in range(1, T):
        clean = [s[0][i] for s in seqs]
        dirty = list(clean)
        dirty[1] = noise
        a, b = ref.track_new(i, clean), bt.track_new(i, dirty)
        for k in (0, 2):
            # neighbours: unaffected (two tracker instances on the same frames differ by at most an ulp of float32 in a corner, 1.5e-5 px
            # observed: the library convolutions inside the estimator are not run-to-run deterministic; a leak from sequence 1 would be pixels)
            np.testing.assert_allclose(a[k]["points"], b[k]["points"], rtol=0, atol=1e-4)
        assert float(np.abs(a[1]["points"] - b[1]["points"]).max()) > 1e-2        # and the fed sequence itself did change
    # the per-sequence scores exist and differ
    assert bt.last_score.shape == (n,) and float(bt.last_score[1]) != float(bt.last_score[0])
    # at n = 1 the batched tracker IS the single tracker
    one = BatchedHomoTracker(net, 1)
    one.init(*_init_args(seqs[:1]))
    solo = HomoTracker(net)
    fr, _, init = seqs[0]
    solo.init(fr[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
    for i in range(1, T):
        np.testing.assert_allclose(one.track_new(i, [fr[i]])[0]["points"], solo.track_new(i, fr[i])["points"], atol=1e-5)


def test_batched_device_tracker_production_shape(dev):
    """BatchedDeviceTracker(model, n) — n x the object install(tracker=True) registers — around the production-shaped stand-in
    (ResNet-50 backbone on PyTorch-ROCm, 256-channel heads: prod29 / circ13 correlations at batch n), one hipGraph per step,
    against the CPU restatement per sequence."""
    import production_standin as PS
    from synth_sequence import success_4pts_error
    from test_gpu_parity import _seeded_net
    from hdn_amd.batched_tracker import BatchedDeviceTracker
    from oracle.tracker_oracle import HomoTrackerOracle, SimilarityOracle
    n, T = 2, 5
    seqs = _sequences(n, T, frame_hw=(480, 854))
    net = _seeded_net()
    net.fc.bias.data.mul_(0.1)
    net_cpu = copy.deepcopy(net)
    twin = PS.ProductionStandIn(net)
    twin.calibrate(*PS.calibration_crops(seqs[0][0], seqs[0][2]))
    cpu_model = PS.ProductionStandInCPU(twin)
    sd = {k: v.clone() for k, v in net_cpu.ShareFeature.state_dict().items()}
    cpu = []
    for frames, _, init in seqs:
        r = HomoTrackerOracle(sd, lambda f: net_cpu.fc(net_cpu.avgpool(net_cpu.backbone(f)).flatten(1)), 1, similarity=SimilarityOracle(cpu_model))
        r.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
        cpu.append([r.track_new(i, frames[i]) for i in range(1, T)])
    model = twin.to(dev).eval()
    bt = BatchedDeviceTracker(model, n)
    assert bt.use_graph is True
    bt.init(*_init_args(seqs))
    assert [tuple(z.shape) for z in model.zf] == [(n, 256, 7, 7)] * 3
    errs = []
    for i in range(1, T):
        res = bt.track_new(i, [s[0][i] for s in seqs])
        errs.append([success_4pts_error(res[b]["points"], cpu[b][i - 1]["points"]) for b in range(n)])
    print("production-shaped batched tracker, corner error vs CPU loop (px):", " ".join(f"{max(e):.1e}" for e in errs))
    assert bt._graph is not None
    assert max(errs[0]) <= 1e-3 and max(max(e) for e in errs) <= 0.1, errs
