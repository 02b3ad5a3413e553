"""ctypes binding of libhdn_hip.so (the C ABI declared in include/hdn_hip.h).

There is no CPU fallback: if the shared library is missing, or a tensor is not on a
ROCm device, the call raises.  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libhdn_hip.so"
LIB_PATH = os.environ.get("HDN_LIB_PATH", os.path.join(_HERE, LIB_NAME))  # override: A/B builds of the kernels
ABI_VERSION = 10

_c_float_p = ctypes.c_void_p  # device pointers travel as integers
_i = ctypes.c_int

# name -> (restype, argtypes); mirrors include/hdn_hip.h one to one
SIGNATURES = {
    "hdn_abi_version": (_i, []),
    "hdn_last_xcorr_variant": (ctypes.c_char_p, []),
    "hdn_xcorr_north_variant": (_i, [_i]),
    "hdn_xcorr_depthwise_f32": (_i, [_c_float_p] * 3 + [_i] * 6 + [ctypes.c_void_p]),
    "hdn_xcorr_depthwise_circ_f32": (_i, [_c_float_p] * 3 + [_i] * 6 + [ctypes.c_void_p]),
    "hdn_xcorr_depthwise_multi_f32": (
        _i,
        [ctypes.POINTER(ctypes.c_void_p)] * 3 + [_i, _i] + [_i] * 6 + [ctypes.c_void_p],
    ),
    "hdn_xcorr_fast_f32": (_i, [_c_float_p] * 3 + [_i] * 7 + [ctypes.c_void_p]),
    "hdn_share_feature_f32": (_i, [_c_float_p] * 3 + [_i] * 3 + [ctypes.c_void_p]),
    "hdn_dlt_solve_f32": (_i, [_c_float_p] * 3 + [_i, ctypes.c_void_p]),
    "hdn_warp_f32": (_i, [_c_float_p] * 3 + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_warp_count_f32": (_i, [_c_float_p] * 4 + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_dlt_warp_f32": (_i, [_c_float_p] * 5 + [_i] * 3 + [ctypes.c_void_p]),
    "hdn_dlt_warp_strided_f32": (_i, [_c_float_p] * 3 + [ctypes.c_longlong] + [_c_float_p] * 2 + [_i] * 3 + [ctypes.c_void_p]),
    "hdn_refine_warp_f32": (_i, [_c_float_p] * 4 + [_i] * 3 + [ctypes.c_void_p]),
    "hdn_l1_score_f32": (_i, [_c_float_p] * 3 + [_i, ctypes.c_float, ctypes.c_void_p]),
    "hdn_l1_score2_f32": (_i, [_c_float_p] * 4 + [_i, ctypes.c_float, ctypes.c_void_p]),
    "hdn_subwindow_f32": (_i, [_c_float_p] * 3 + [_i] * 5 + [ctypes.c_void_p]),
    "hdn_frame_warp_perspective_u8": (_i, [_c_float_p] * 3 + [_i] * 3 + [ctypes.c_void_p]),
    "hdn_frame_warp_affine_cubic_u8": (_i, [_c_float_p] * 3 + [_i] * 3 + [ctypes.c_void_p]),
    "hdn_subwindow_batch_f32": (_i, [_c_float_p] * 2 + [_i, _c_float_p] + [_i] * 6 + [ctypes.c_void_p]),
    "hdn_frame_warp_perspective_batch_u8": (_i, [_c_float_p] * 2 + [_i, _c_float_p] + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_frame_warp_affine_cubic_batch_u8": (_i, [_c_float_p] * 2 + [_i, _c_float_p] + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_l1_score2_batch_f32": (_i, [_c_float_p] * 4 + [_i, ctypes.c_longlong, _i, ctypes.c_float, ctypes.c_void_p]),
    "hdn_remap_linear_f32": (_i, [_c_float_p] * 4 + [_i] * 5 + [ctypes.c_void_p]),
    "hdn_similarity_translation_f32": (_i, [_c_float_p] * 6 + [_i, _i, ctypes.c_double, ctypes.c_float, ctypes.c_double, _i, ctypes.c_void_p]),
    "hdn_similarity_logpolar_f32": (_i, [_c_float_p] * 5 + [_i, _i, ctypes.c_float, ctypes.c_double, ctypes.c_float, _i, ctypes.c_void_p]),
    "hdn_simi_track_update_f64": (_i, [_c_float_p] * 4 + [_i, _i, _i, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]),
    "hdn_track_prepare_f64": (_i, [_c_float_p] * 3 + [_i, ctypes.c_void_p]),
    "hdn_track_accumulate_f64": (_i, [_c_float_p] * 6 + [_i] + [_c_float_p] * 2 + [_i, ctypes.c_void_p]),
    "hdn_trunk_stem_f32": (_i, [_c_float_p] * 4 + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_conv3x3s2_v2_f32": (_i, [_c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p] + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_trunk_stem_mfma_f32": (_i, [_c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p] + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_bias_relu_f32": (_i, [_c_float_p] * 3 + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_avgpool_fc_f32": (_i, [_c_float_p] * 4 + [_i] * 6 + [ctypes.c_void_p]),
    "hdn_head_tail_f32": (_i, [_c_float_p, ctypes.c_void_p] + [_c_float_p] * 4 + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_head_conv3x3_f32": (_i, [ctypes.c_void_p, ctypes.c_void_p, _c_float_p, ctypes.c_void_p] + [_i] * 5 + [ctypes.c_void_p]),
    "hdn_set_check_range": (_i, [_i]),
    "hdn_xcorr_north_launch_events": (_i, [ctypes.c_void_p, ctypes.c_void_p]),
    "hdn_pack_conv3x3_bytes": (ctypes.c_longlong, [_i]),
    "hdn_pack_conv3x3_f32": (_i, [_c_float_p, _i, ctypes.c_void_p, ctypes.c_longlong]),
    "hdn_pack_conv3x3s2_ds_bytes": (ctypes.c_longlong, [_i]),
    "hdn_pack_conv3x3s2_ds_f32": (_i, [_c_float_p, _c_float_p, _i, ctypes.c_void_p, ctypes.c_longlong]),
    "hdn_pack_conv3x3_v2_bytes": (ctypes.c_longlong, [_i]),
    "hdn_pack_conv3x3_v2_f32": (_i, [_c_float_p, _i, ctypes.c_void_p, ctypes.c_longlong]),
    "hdn_pack_conv3x3s2_v2_bytes": (ctypes.c_longlong, [_i]),
    "hdn_pack_conv3x3s2_v2_f32": (_i, [_c_float_p, _c_float_p, _i, ctypes.c_void_p, ctypes.c_longlong]),
    "hdn_pack_stem_mfma_bytes": (ctypes.c_longlong, []),
    "hdn_pack_stem_mfma_f32": (_i, [_c_float_p, ctypes.c_void_p, ctypes.c_longlong]),
    "hdn_pack_head_conv3x3_bytes": (ctypes.c_longlong, [_i, _i]),
    "hdn_pack_head_conv3x3_f32": (_i, [ctypes.c_void_p, _i, _i, ctypes.c_void_p, ctypes.c_longlong]),
    "hdn_pack_head_tail_bytes": (ctypes.c_longlong, [_i, _i]),
    "hdn_pack_head_tail_f32": (_i, [_c_float_p, _i, _i, ctypes.c_void_p, ctypes.c_longlong]),
    "hdn_ubench_copy_f32": (_i, [_c_float_p] * 2 + [ctypes.c_longlong, ctypes.c_void_p]),
    "hdn_conv3x3_pack_info": (_i, [_i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "hdn_conv3x3_workspace_bytes": (ctypes.c_longlong, [_i, _i, _i, _i]),
    "hdn_conv3x3s2_ds_f32": (_i, [_c_float_p] * 6 + [ctypes.c_longlong] + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_conv3x3_bias_relu_f32": (_i, [_c_float_p] * 6 + [ctypes.c_longlong] + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_conv3x3_v2_pack_info": (_i, [_i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "hdn_conv3x3_v2_workspace_bytes": (ctypes.c_longlong, [_i, _i, _i]),
    "hdn_conv3x3_v2_f32": (_i, [_c_float_p] * 6 + [ctypes.c_longlong] + [_i] * 4 + [ctypes.c_void_p]),
    "hdn_conv3x3_chain_slices": (_i, [_i, _i, _i, _i]),
    "hdn_conv3x3_chain_f32": (_i, [_c_float_p, _i, _c_float_p, _c_float_p, _i, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p] + [_i] * 5 + [ctypes.c_void_p]),
    "hdn_conv3x3_finish_f32": (_i, [_c_float_p, _i, _c_float_p, _c_float_p, _i, _c_float_p] + [_i] * 3 + [ctypes.c_void_p]),
    "hdn_allgather_offsets": (_i, [_c_float_p] * 2 + [_i, ctypes.c_void_p, ctypes.c_void_p]),
    "hdn_rccl_available": (_i, []),
    "hdn_rccl_unique_id": (_i, [ctypes.c_void_p]),
    "hdn_rccl_comm_create": (_i, [ctypes.POINTER(ctypes.c_void_p), _i, _i, ctypes.c_void_p]),
    "hdn_rccl_comm_count": (_i, [ctypes.c_void_p, ctypes.POINTER(_i)]),
    "hdn_rccl_comm_destroy": (_i, [ctypes.c_void_p]),
    "hdn_gather_create": (_i, [ctypes.POINTER(ctypes.c_void_p), _i, _i, ctypes.c_longlong]),
    "hdn_gather_handle": (_i, [ctypes.c_void_p, ctypes.c_void_p]),
    "hdn_gather_connect": (_i, [ctypes.c_void_p, ctypes.c_void_p]),
    "hdn_gather_offsets_oneshot": (_i, [ctypes.c_void_p, _c_float_p, _c_float_p, _i, ctypes.c_void_p]),
    "hdn_gather_status": (_i, [ctypes.c_void_p]),
    "hdn_gather_destroy": (_i, [ctypes.c_void_p]),
    "hdn_gather_peer_access": (_i, [_i, ctypes.c_char_p]),
    "hdn_device_pci_bus_id": (_i, [_i, ctypes.c_char_p, _i]),
    "hdn_logpolar_sample_f32": (_i, [_c_float_p] * 7 + [_i] * 5 + [ctypes.c_void_p]),
}

ERRORS = {
    -1: "HDN_E_NULL: a required pointer is NULL",
    -2: "HDN_E_SHAPE: non-positive size, or correlation kernel larger than the search plane",
    -3: "HDN_E_LIMIT: size exceeds what the kernels support",
    -4: "HDN_E_ALIAS: output aliases an input",
    -5: "HDN_E_NORCCL: librccl.so.1 could not be loaded",
    -6: "HDN_E_PEER: an earlier one-shot gather timed out waiting for a peer; the context is unusable",
}

_lock = threading.Lock()
_lib = None


class HdnHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libhdn_hip.so (built in-tree by __graft_entry__.build()).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HdnHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  hdn_amd has no CPU fallback."
            )
        # torch is imported first so that the HIP runtime torch bundles (SONAME libamdhip64.so.7) is the one
        # this library binds to: device pointers and streams are then shared with torch.
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        got = lib.hdn_abi_version()
        if got != ABI_VERSION:
            raise HdnHipError(f"{LIB_NAME} ABI version {got} != expected {ABI_VERSION}; rebuild")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc in ERRORS:
        raise ValueError(f"{what}: {ERRORS[rc]}")
    if rc <= -2000:
        raise HdnHipError(f"{what}: RCCL call failed with ncclResult_t {-rc - 2000}")
    if rc <= -1000:
        raise HdnHipError(f"{what}: HIP launch failed with hipError_t {-rc - 1000}")
    raise HdnHipError(f"{what}: unknown error code {rc}")


def require_device(*tensors: torch.Tensor) -> torch.device:
    """All tensors must be fp32 tensors on one ROCm device.  No silent CPU path."""
    dev = None
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected a torch.Tensor, got {type(t).__name__}")
        if not t.is_cuda:
            raise HdnHipError(
                "hdn_amd runs on the GPU only (tensor is on %s); there is no CPU fallback" % t.device
            )
        if t.dtype != torch.float32:
            raise TypeError(f"hdn_amd kernels compute in fp32; got {t.dtype}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise HdnHipError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device: torch.device) -> ctypes.c_void_p:
    """The hipStream_t torch is currently issuing work on for `device` (the raw handle: no Stream object per launch)."""
    if _raw_stream is not None:
        idx = device.index
        return ctypes.c_void_p(_raw_stream(idx if idx is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device: torch.device):
    """`with device_guard(dev):` = `with torch.cuda.device(dev):`, free when `dev` already is the current device (the one-process-per-GPU
    case: a launch wrapper should not pay two device switches of bookkeeping per kernel)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def ptr(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr())
