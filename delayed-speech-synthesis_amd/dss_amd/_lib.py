"""ctypes binding of libdss_hip.so (the C ABI in include/dss_hip.h).

torch is imported first on purpose: PyTorch-ROCm ships its own libamdhip64.so (soname libamdhip64.so.7);
loading it before our library makes both share ONE HIP runtime, so torch device pointers and streams can
be handed to the C ABI.  There is no CPU fallback: if the library is missing it is built with hipcc, and
if no GPU is present every compute call raises DssError.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_lib = None


class DssError(RuntimeError):
    pass


def _declare(L):
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    sig = {
        "dss_last_error": (C.c_char_p, []),
        "dss_version": (C.c_char_p, []),
        "dss_device_count": (i, []),
        "dss_set_device": (i, [i]),
        "dss_current_device": (i, []),
        "dss_stream_create": (vp, []),
        "dss_stream_destroy": (None, [vp]),
        "dss_stream_synchronize": (i, [vp]),
        "dss_event_create": (vp, []),
        "dss_event_destroy": (None, [vp]),
        "dss_event_record": (i, [vp, vp]),
        "dss_event_query": (i, [vp]),
        "dss_event_synchronize": (i, [vp]),
        "dss_stream_wait_event": (i, [vp, vp]),
        "dss_host_alloc": (vp, [sz, i]),
        "dss_host_free": (None, [vp]),
        "dss_memcpy_d2h_async": (i, [vp, vp, sz, vp]),
        "lpcnet_create": (vp, []),
        "lpcnet_init": (i, [vp]),
        "lpcnet_destroy": (None, [vp]),
        "lpcnet_synthesize": (None, [vp, vp, vp, i]),
        "lpcnet_get_size": (i, []),
        "dss_error_count": (C.c_long, []),
        "lpcnet_encoder_create": (vp, []),
        "lpcnet_encoder_init": (i, [vp]),
        "lpcnet_encoder_destroy": (None, [vp]),
        "lpcnet_compute_features": (i, [vp, vp, vp]),
        "lpcnet_compute_single_frame_features": (i, [vp, vp, vp]),
        "dss_lpcnet_model_info": (i, [vp, vp, vp, vp, vp]),
        "dss_lpcnet_batch_force_excitation": (i, [vp, vp, i, i]),
        "dss_selftest_exp10": (i, [vp, vp, vp, C.c_long]),
        "dss_selftest_lin2ulaw": (i, [C.c_uint, C.c_uint, C.c_long, vp]),
        "dss_selftest_fast_layout": (i, [vp, C.c_size_t, vp]),
        "dss_lpcnet_load_model": (i, [C.c_char_p, sz]),
        "dss_lpcnet_load_model_file": (i, [C.c_char_p]),
        "dss_lpcnet_bytes_per_sample": (C.c_double, []),
        "dss_lpcnet_batch_create": (vp, [i, i]),
        "dss_lpcnet_batch_destroy": (None, [vp]),
        "dss_lpcnet_batch_create_lane": (vp, [vp, i, i]),
        "dss_lpcnet_batch_reset": (i, [vp, i]),
        "dss_lpcnet_batch_reset_async": (i, [vp, i, vp]),
        "dss_lpcnet_batch_synthesize": (i, [vp, vp, i, i, i, vp]),
        "dss_lpcnet_batch_synthesize_dev": (i, [vp, vp, i, i, i, vp, vp]),
        "dss_lpcnet_batch_synthesize_ragged": (i, [vp, vp, vp, vp, i, i, i, vp]),
        "dss_lpcnet_batch_synthesize_ragged_dev": (i, [vp, vp, vp, vp, i, i, i, vp, vp]),
        "dss_lpcnet_batch_tap": (i, [vp, i, i, vp, sz]),
        "dss_lpcnet_batch_enable_trace": (i, [vp, i]),
        "dss_lpcnet_batch_set_multi": (i, [vp, i]),
        "dss_lpcnet_batch_enable_timing": (i, [vp, i]),
        "dss_lpcnet_batch_kernel_ms": (C.c_double, [vp, i]),
        "dss_gate_create": (vp, [i, i, i, C.c_double, i, i, i]),
        "dss_gate_destroy": (None, [vp]),
        "dss_gate_reset": (i, [vp, i]),
        "dss_gate_max_events": (i, [vp]),
        "dss_gate_push": (i, [vp, vp, vp, i, vp]),
        "dss_gate_push_dev": (i, [vp, vp, vp, i, vp, vp]),
        "dss_gate_segment": (i, [vp, i, i, vp, i]),
        "dss_gate_segment_dev": (i, [vp, i, i, vp, i, vp]),
        "dss_gate_collect_dev": (i, [vp, i, vp, vp, vp, vp, i, vp]),
        "dss_gate_frames_seen": (i, [vp, i]),
        "dss_vad_create": (vp, [i, i, i]),
        "dss_vad_destroy": (None, [vp]),
        "dss_vad_load_weights": (i, [vp] * 11),
        "dss_vad_reset": (i, [vp, i]),
        "dss_vad_reset_async": (i, [vp, i, vp]),
        "dss_vad_step_dev": (i, [vp, vp, i, i, vp, vp, vp]),
        "dss_vad_state": (i, [vp, vp, vp, i]),
        "dss_dec_create": (vp, [i, i, i, i, i]),
        "dss_dec_destroy": (None, [vp]),
        "dss_dec_load_weights": (i, [vp, vp]),
        "dss_dec_forward_dev": (i, [vp, vp, i, i, i, vp, vp]),
        "dss_dec_forward_rows_dev": (i, [vp, vp, i, i, vp, vp, i, i, vp, vp]),
        "dss_hga_num_windows": (i, [i, i, f, f]),
        "dss_hga_log_power": (i, [vp, i, i, i, f, f, vp]),
        "dss_hga_create": (vp, [i, i, i, f, f, i, vp, vp, vp, vp]),
        "dss_hga_destroy": (None, [vp]),
        "dss_hga_reset": (i, [vp]),
        "dss_hga_frames_for": (i, [vp, i]),
        "dss_hga_extract": (i, [vp, vp, i, vp]),
        "dss_hga_extract_dev": (i, [vp, vp, i, vp, i, vp]),
        "dss_hga_set_frontend": (i, [vp, i, vp, vp, i, vp, vp]),
        "dss_hga_extract_raw": (i, [vp, vp, i, vp]),
        "dss_hga_extract_raw_dev": (i, [vp, vp, i, vp, i, vp]),
        "dss_hga_extract_wire_dev": (i, [vp, vp, i, vp, i, vp]),
        "dss_hga_set_zscore": (i, [vp, vp, vp]),
        "dss_selftest_hga_force_path": (i, [vp, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return sig


EXPORTED_SYMBOLS = None


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True):
    """Load (building first if needed) libdss_hip.so.  Raises if it cannot be produced."""
    global _lib, EXPORTED_SYMBOLS
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  -- share torch's HIP runtime (see module docstring)
    except Exception:  # pragma: no cover - torch is part of the image
        pass
    path = _build.LIB_PATH
    if build_if_missing and _build.needs_build():
        _build.build_library()
    if not os.path.exists(path):
        raise DssError(f"{path} is missing and could not be built; there is no CPU fallback")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    EXPORTED_SYMBOLS = sorted(_declare(L))
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise DssError(f"libdss_hip error {rc}: {load().dss_last_error().decode()}")
    return rc


def require_gpu():
    L = load()
    if L.dss_device_count() <= 0:
        raise DssError("no HIP device visible: libdss_hip has no CPU fallback")
    return L
