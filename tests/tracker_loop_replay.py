"""TEST INFRASTRUCTURE shared by tests/test_oracle_golden.py (CPU) and tests/test_gpu_tracker.py (-m gpu): replay of the network outputs
recorded in tests/golden/tracker_loop.npz — the fixture make_golden.py:gen_tracker_loop wrote while the REFERENCE's own
hdnTrackerHomo.init / track_new (hdn/tracker/hdn_tracker_proj_e2e.py:60-120,141-285) executed around its real ModelBuilder — so that a
tracker loop under test (the oracle's, or the device-resident one) can be driven with exactly the maps / offsets the reference's loop saw
and compared with what the reference computed from them."""
import numpy as np
import pytest
import torch

from conftest import load_golden  # noqa: F401


class ReplayModel:
    """Plays back the network outputs tests/golden/tracker_loop.npz recorded while the reference's own hdnTrackerHomo.init / track_new
    ran around its real ModelBuilder: template / track_new / track_new_lp / track_proj (model_builder_e2e_unconstrained_v2.py:87-217)
    return what the reference's networks returned for that frame, and keep what they were handed (the crops) for the caller to check."""

    def __init__(self, g, prefix, to=lambda a: torch.from_numpy(np.ascontiguousarray(a))):
        self.g, self.p, self.to, self.frame, self.seen = g, prefix, to, 0, {}

    def key(self, name):
        return f"{self.p}f{self.frame}__{name}"

    def template(self, z):
        self.seen["z_crop"] = z

    def track_new(self, x, delta=[0, 0]):
        self.seen["x_crop"] = x
        return {"cls": self.to(self.g[self.key("cls")]), "loc_c": self.to(self.g[self.key("loc_c")])}

    def track_new_lp(self, x, delta=[0, 0]):
        self.seen["x_crop_moved"] = x
        return {"cls_lp": self.to(self.g[self.key("cls_lp")]), "loc_lp": self.to(self.g[self.key("loc_lp")])}

    def track_proj(self, tmpl, srch):
        self.seen["search"] = srch
        return self.to(self.g[self.key("H_mat")]), self.to(self.g[self.key("homo_score")]), self.to(self.g[self.key("simi_score")])


def crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def tracker_loop_sequence(g):
    """The frames of tests/golden/tracker_loop.npz, regenerated from the seed (tools/synth_sequence.py) and checked against the
    fixture's CRCs: a NumPy whose FFT / generator differs from the build container's is reported as such, not as a parity failure."""
    from tools.synth_sequence import make_sequence
    frames, corners, init = make_sequence(n_frames=int(g["seq__n_frames"]), frame_hw=tuple(int(v) for v in g["seq__frame_hw"]),
                                          target_wh=tuple(int(v) for v in g["seq__target_wh"]), seed=int(g["seq__seed"]))
    got = np.array([crc(f) for f in frames], np.int64)
    if not np.array_equal(got, g["seq__frames_crc"]):
        pytest.skip("tools/synth_sequence.make_sequence does not reproduce the fixture's frames on this NumPy build")
    np.testing.assert_array_equal(np.array(init["poly"]), g["seq__poly"])
    return frames, init
