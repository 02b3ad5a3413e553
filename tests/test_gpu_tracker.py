"""BASELINE configs[3] harness (-m gpu): a seeded synthetic planar-target sequence streamed through the device-resident
tracker loop (hdn_amd.tracker.HomoTracker: one frame upload, kernels, one host read per frame) against the CPU restatement of
the same loop (oracle/tracker_oracle.py), frame by frame, with the reference's corner-error metric
(success_4pts_error, toolkit/utils/statistics.py:206-218).  POT-210 and OpenCV are not available, the head's weights are
seeded, so what is measured is GPU-vs-CPU agreement of the whole per-frame chain, not tracking accuracy."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_sequence_stream_device_loop_vs_cpu_restatement(dev):
    from synth_sequence import make_sequence, success_4pts_error
    from test_gpu_parity import _seeded_net
    from hdn_amd.tracker import HomoTracker
    from oracle.tracker_oracle import HomoTrackerOracle
    frames, corners, init = make_sequence(n_frames=16, frame_hw=(360, 640), target_wh=(150, 100), seed=7)
    net = _seeded_net()
    net_cpu = copy.deepcopy(net)
    sd = {k: v.clone() for k, v in net_cpu.ShareFeature.state_dict().items()}
    ref = HomoTrackerOracle(sd, lambda f: net_cpu.fc(net_cpu.avgpool(net_cpu.backbone(f)).flatten(1)))
    trk = HomoTracker(net.to(dev))
    ref.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
    trk.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
    np.testing.assert_allclose(trk.channel_average, ref.channel_average, rtol=1e-12)
    assert trk.z_crop_points_sm == tuple(ref.z_crop_points_sm)
    np.testing.assert_allclose(trk.init_homo_tmp.cpu().numpy()[0], ref.init_homo_tmp.astype(np.float32), atol=1e-6)
    errs = []
    for t in range(1, len(frames)):
        a = trk.track_new(t, frames[t])
        b = ref.track_new(t, frames[t])
        assert set(a) == {"bbox_aligned", "best_score", "polygon", "points", "bbox"} and a["points"].shape == (4, 2)
        errs.append(success_4pts_error(a["points"], b["points"]))
    # the chain is a recurrence (H_total feeds the next frame's warp, whose 1/32-px taps feed the head): differences of
    # ~1e-5 px per frame in the head's offsets may grow, but stay far below a pixel over the sequence
    assert errs[0] <= 1e-3, errs
    assert max(errs[:5]) <= 1e-2 and max(errs) <= 0.25, errs
    assert trk.host_syncs == 1 + (len(frames) - 1)   # the channel mean at init + one read of 4 corners per frame
    # without the host read nothing synchronises: the same frame again, asynchronously, gives device tensors
    out = trk.track_new(99, frames[-1], sync=False)
    assert out["points"].is_cuda and trk.host_syncs == len(frames)


def test_graphed_tracker_loop_matches_eager(dev):
    """The whole per-frame body as one hipGraph replay (static frame buffer, H_total carried inside the graph) gives the
    eager loop's corners."""
    from synth_sequence import make_sequence, success_4pts_error
    from test_gpu_parity import _seeded_net
    from hdn_amd.tracker import HomoTracker
    frames, corners, init = make_sequence(n_frames=8, frame_hw=(360, 640), target_wh=(150, 100), seed=8)
    net = _seeded_net().to(dev)
    a, b = HomoTracker(net), HomoTracker(net, graph=True)
    for t in (a, b):
        t.init(frames[0], init["bbox"], init["poly"], init["gt_points"], init["first_point"])
    for i in range(1, len(frames)):
        pa, pb = a.track_new(i, frames[i])["points"], b.track_new(i, frames[i])["points"]
        # (the seeded head does not track: H_total drifts by ~10 % per frame, and with it the sensitivity to the last bits of
        #  the closed-form inverse the captured body uses; observed 0 ... 4e-5 px over the first 5 frames)
        assert success_4pts_error(pa, pb) <= (1e-3 if i <= 3 else 2e-2), (i, pa, pb)
    assert b._graph is not None
    np.testing.assert_allclose(a.H_total.cpu().numpy(), b.H_total.cpu().numpy(), rtol=1e-4, atol=1e-3)
