"""hdn_amd — MI355X-native implementation of HDN's per-frame homography-estimation hot path.

Hand-written gfx950 HIP kernels behind a C ABI (include/hdn_hip.h, hdn_amd/libhdn_hip.so), exposed under
the reference's own Python signatures.  See DESIGN.md and INTEGRATION.md.
"""
from .xcorr import xcorr_depthwise, xcorr_depthwise_circular, xcorr_depthwise_multi, xcorr_fast, xcorr_slow  # noqa: F401
from .homography import DLT_solve, transform, transformer, Homo_STN, dlt_warp  # noqa: F401
from .share_feature import PreShareFeature, fold_params  # noqa: F401
from .homo_model import HomoModelBuilder, track_proj  # noqa: F401
from .logpolar import STN_Polar  # noqa: F401
from .heads import MultiBAN, MultiCircBAN  # noqa: F401
from .refine import homo_refine, refine_warp  # noqa: F401

__version__ = "0.1.0"
from .backbone import optimize_similarity_model, restore_similarity_model  # noqa: F401
