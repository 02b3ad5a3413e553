import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 20260928  # same seed as tests/golden/make_golden.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_rng(tag):
    return np.random.default_rng(SEED + tag)


def relu_normal(g, shape):
    return np.maximum(g.standard_normal(shape, dtype=np.float32), 0.0)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def seeded_bn_(bn, g):
    """Same as tests/golden/make_golden.py:seeded_bn_ (non-trivial eval-mode BN statistics)."""
    import torch
    n = bn.num_features
    for name, lo, hi in (("weight", 0.5, 1.5), ("bias", -0.3, 0.3), ("running_mean", -0.5, 0.5), ("running_var", 0.5, 2.0)):
        getattr(bn, name).data = torch.from_numpy(g.uniform(lo, hi, n).astype(np.float32))


def seeded_head256(tag):
    """The 256-channel MultiBAN / MultiCircBAN of tests/golden/heads256.npz, re-created from the seed exactly as
    make_golden.py:gen_heads creates the reference's (same constructor order => same torch init stream; the
    fixture's `param_sum` detects a drift).  Returns (module, z_fs, x_fs)."""
    import torch
    from hdn_amd import heads
    # tag "ban_cfg5": heads256_cfg5.npz, the MultiBAN of BASELINE configs[4] (37 x 37 search features -> 31 x 31 maps)
    cls, zsz, xsz, seed, gtag = {"ban": (heads.MultiBAN, 7, 31, 13, 810), "circ": (heads.MultiCircBAN, 15, 15, 14, 811),
                                 "ban_cfg5": (heads.MultiBAN, 7, 37, 15, 812)}[tag]
    torch.manual_seed(SEED + seed)
    m = cls([256, 256, 256], 2, weighted=True).eval()
    g = golden_rng(gtag)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            seeded_bn_(mod, g)
    m.cls_weight.data = torch.from_numpy(g.standard_normal(3).astype(np.float32))
    m.loc_weight.data = torch.from_numpy(g.standard_normal(3).astype(np.float32))
    m.loc_scale.data = torch.from_numpy(g.uniform(0.5, 1.5, 3).astype(np.float32))
    zfs = [torch.from_numpy(g.standard_normal((1, 256, zsz, zsz), dtype=np.float32)) for _ in range(3)]
    xfs = [torch.from_numpy(g.standard_normal((1, 256, xsz, xsz), dtype=np.float32)) for _ in range(3)]
    return m, zfs, xfs


@pytest.fixture(scope="session", autouse=True)
def _fp16_piece_range_guard_on_for_gpu_tests():
    """The -m gpu suite runs with the fp16-piece range guard on (include/hdn_hip.h, hdn_set_check_range): every convolution /
    head-kernel call first checks max |input| < 65,504 and would fail with HDN_E_LIMIT instead of answering inf / NaN."""
    try:
        import torch
        if not torch.cuda.is_available():
            yield
            return
        from hdn_amd import _lib
        lib = _lib.load()
    except Exception:
        yield
        return
    prev = lib.hdn_set_check_range(1)
    yield
    lib.hdn_set_check_range(prev)
