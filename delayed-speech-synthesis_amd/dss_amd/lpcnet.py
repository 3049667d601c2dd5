"""Batched LPCNet vocoder on MI355X: Python host side of Part 1/2 of include/dss_hip.h.

``LPCNetBatch`` is the GPU replacement of the reference's only bulk use of the vocoder,
``AsynchronousSynthesisQueue`` (local/training.py:165-207: one process and one fresh ``LPCNet()`` per
utterance), and of the per-row loop in ``DelayedLPCNetVocoder.synthesize`` (local/units.py:531-538).
Each of the B slots is an independent persistent decoder state, exactly like B ``LPCNet.LPCNet`` objects.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .lpcnet_weights import synthetic_blob

FRAME_SIZE = 160
NB_FEATURES = 20

_model_loaded = False


def load_model(blob: Optional[bytes] = None, path: Optional[str] = None) -> None:
    """Make a weight blob the process-wide model (what lpcnet_create()/LPCNet() bind to)."""
    global _model_loaded
    L = _lib.load()
    if path is not None:
        _lib.check(L.dss_lpcnet_load_model_file(path.encode()))
    else:
        if blob is None:
            blob = synthetic_blob(0)
        _lib.check(L.dss_lpcnet_load_model(blob, len(blob)))
    _model_loaded = True


def ensure_model() -> None:
    """Load $DSS_LPCNET_WEIGHTS if set, else the seeded synthetic model (real xiph weights cannot be
    fetched offline)."""
    import os
    if _model_loaded:
        return
    p = os.environ.get("DSS_LPCNET_WEIGHTS")
    load_model(path=p) if p else load_model()


def bytes_per_sample() -> float:
    return float(_lib.load().dss_lpcnet_bytes_per_sample())


class LPCNetBatch:
    """B persistent decoder states on one GPU."""

    def __init__(self, max_utts: int, max_frames: int):
        L = _lib.require_gpu()
        ensure_model()
        self._L = L
        self.max_utts, self.max_frames = int(max_utts), int(max_frames)
        self._h = L.dss_lpcnet_batch_create(self.max_utts, self.max_frames)
        if not self._h:
            raise _lib.DssError(L.dss_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.dss_lpcnet_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self, utt: int = -1):
        _lib.check(self._L.dss_lpcnet_batch_reset(self._h, int(utt)))

    def reset_async(self, utt: int = -1, stream=None):
        import torch
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _lib.check(self._L.dss_lpcnet_batch_reset_async(self._h, int(utt), s))

    def synthesize(self, features: np.ndarray) -> np.ndarray:
        """features (B, F, >=20) float32 host array -> (B, F*160) int16.  State carries to the next call."""
        f = np.ascontiguousarray(features, dtype=np.float32)
        if f.ndim != 3 or f.shape[2] < NB_FEATURES:
            raise ValueError("features must be (B, F, >=20)")
        B, F, S = f.shape
        pcm = np.empty((B, F * FRAME_SIZE), dtype=np.int16)
        _lib.check(self._L.dss_lpcnet_batch_synthesize(self._h, f.ctypes.data, B, F, S, pcm.ctypes.data))
        return pcm

    def synthesize_torch(self, features, out=None, stream=None):
        """Device-resident form: features is a CUDA (HIP) float32 tensor (B, F, S>=20); returns an int16 CUDA
        tensor (B, F*160).  Asynchronous on the current torch stream."""
        import torch
        assert features.is_cuda and features.dtype == torch.float32 and features.is_contiguous()
        B, F, S = features.shape
        if out is None:
            out = torch.empty((B, F * FRAME_SIZE), dtype=torch.int16, device=features.device)
        s = torch.cuda.current_stream(features.device).cuda_stream if stream is None else stream
        _lib.check(self._L.dss_lpcnet_batch_synthesize_dev(self._h, features.data_ptr(), B, F, S, out.data_ptr(), s))
        return out

    # ---- test / measurement taps ------------------------------------------------------------------------
    def enable_trace(self, on=True):
        _lib.check(self._L.dss_lpcnet_batch_enable_trace(self._h, int(on)))

    def enable_timing(self, on=True):
        _lib.check(self._L.dss_lpcnet_batch_enable_timing(self._h, int(on)))

    def kernel_ms(self, which=0) -> float:
        return float(self._L.dss_lpcnet_batch_kernel_ms(self._h, which))

    def tap(self, utt: int, which: int, n_frames: int) -> np.ndarray:
        width = {0: 1152, 1: 48, 2: 16, 3: FRAME_SIZE, 4: FRAME_SIZE}[which]
        out = np.empty((n_frames, width), dtype=np.float32)
        _lib.check(self._L.dss_lpcnet_batch_tap(self._h, utt, which, out.ctypes.data, out.size))
        return out
