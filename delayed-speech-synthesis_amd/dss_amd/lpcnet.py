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


class NoModelError(_lib.DssError):
    pass


def load_model(blob: Optional[bytes] = None, path: Optional[str] = None, synthetic: bool = False) -> None:
    """Make a weight blob the process-wide model (what lpcnet_create()/LPCNet() bind to).

    Exactly one of ``blob`` / ``path`` / ``synthetic=True``.  The synthetic model (seeded random weights of the
    published architecture) exists for tests and benchmarks only -- it produces noise, not speech -- so it is never
    chosen implicitly."""
    global _model_loaded
    L = _lib.load()
    if sum(x is not None for x in (blob, path)) + bool(synthetic) != 1:
        raise ValueError("load_model needs exactly one of blob=, path= or synthetic=True")
    if path is not None:
        _lib.check(L.dss_lpcnet_load_model_file(path.encode()))
    else:
        if synthetic:
            blob = synthetic_blob(0)
        _lib.check(L.dss_lpcnet_load_model(blob, len(blob)))
    _model_loaded = True


def ensure_model() -> None:
    """Bind the process-wide model: an explicitly loaded one, else the file named by $DSS_LPCNET_WEIGHTS, else -- ONLY
    when $DSS_LPCNET_SYNTHETIC=1 (tests, bench.py) -- the seeded synthetic model.  Without any of them this raises:
    a prosthesis must never play noise from random weights because a path was forgotten (the C ABI fails the same
    way, DSS_ENOMODEL)."""
    import os
    if _model_loaded:
        return
    p = os.environ.get("DSS_LPCNET_WEIGHTS")
    if p:
        load_model(path=p)
    elif os.environ.get("DSS_LPCNET_SYNTHETIC") == "1":
        load_model(synthetic=True)
    else:
        raise NoModelError("no LPCNet weights: set DSS_LPCNET_WEIGHTS to a blob converted with `python -m "
                           "dss_amd.nnet_data nnet_data.c model.blob`, or call dss_amd.lpcnet.load_model(); "
                           "DSS_LPCNET_SYNTHETIC=1 selects random test weights (noise, not speech)")


def model_info() -> dict:
    """Which sample-rate kernel the loaded model runs on (``dss_lpcnet_model_info``)."""
    import ctypes
    L = _lib.require_gpu()
    v = [ctypes.c_int(0) for _ in range(5)]
    _lib.check(L.dss_lpcnet_model_info(*[ctypes.byref(x) for x in v]))
    keys = ("fast_path", "zr_slots_max", "h_slots_max", "h_lds_bytes", "gru_a_order")
    info = {k: int(x.value) for k, x in zip(keys, v)}
    info["kernel"] = "lpcnet_sample_kernel" if info["fast_path"] else "lpcnet_sample_generic_kernel"
    info["extended"] = info["fast_path"] == 2      # z/r tail blocks or long h lists in LDS (skewed sparsity)
    return info


def bytes_per_sample() -> float:
    return float(_lib.load().dss_lpcnet_bytes_per_sample())


class LPCNetBatch:
    """B persistent decoder states on one GPU."""

    def __init__(self, max_utts: int, max_frames: int, device: Optional[int] = None):
        L = _lib.require_gpu()
        if device is not None:
            _lib.check(L.dss_set_device(int(device)))
        ensure_model()
        self._L = L
        self.max_utts, self.max_frames = int(max_utts), int(max_frames)
        self._h = L.dss_lpcnet_batch_create(self.max_utts, self.max_frames)
        if not self._h:
            raise _lib.DssError(L.dss_last_error().decode())
        self.device = int(L.dss_current_device())
        info = model_info()
        if not info["fast_path"]:
            import warnings
            warnings.warn("LPCNet model exceeds the CU-resident kernel's capacities (z/r blocks per row group %d of 12+16, "
                          "h blocks %d of 64, LDS image %d of 151552 B): running on the generic kernel, several times slower"
                          % (info["zr_slots_max"], info["h_slots_max"], info["h_lds_bytes"]), RuntimeWarning, stacklevel=2)

    def create_lane(self, max_rows: int, max_frames: int) -> "LPCNetBatch":
        """A second launch context on THIS batch's decoder slots (``dss_lpcnet_batch_create_lane``): it owns per-call scratch
        for max_rows x max_frames, its ragged calls name this batch's slots.  Calls on different lanes may be in flight on
        different streams at once as long as no slot is in two of them (the caller orders a slot's calls)."""
        L = self._L
        _lib.check(L.dss_set_device(self.device))
        h = L.dss_lpcnet_batch_create_lane(self._h, int(max_rows), int(max_frames))
        if not h:
            raise _lib.DssError(L.dss_last_error().decode())
        lane = object.__new__(LPCNetBatch)
        lane._L, lane._h, lane.device = L, h, self.device
        lane.max_utts, lane.max_frames = int(max_rows), int(max_frames)
        lane._parent = self                      # keeps the slots alive as long as the lane
        return lane

    def _check_device(self, t):
        """Kernels of this batch run on self.device; a tensor from another GPU would be a silent peer access."""
        if t.device.index != self.device:
            raise ValueError(f"tensor on cuda:{t.device.index}, decoder batch on cuda:{self.device}")

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
        self._check_device(features)
        B, F, S = features.shape
        if out is None:
            out = torch.empty((B, F * FRAME_SIZE), dtype=torch.int16, device=features.device)
        else:
            self._check_device(out)
        s = torch.cuda.current_stream(features.device).cuda_stream if stream is None else stream
        _lib.check(self._L.dss_lpcnet_batch_synthesize_dev(self._h, features.data_ptr(), B, F, S, out.data_ptr(), s))
        return out

    def synthesize_ragged(self, rows, slots=None, longest_first: bool = True):
        """Utterances of different lengths in one launch.  ``rows``: sequence of (F_i, >=20) float32 host arrays;
        ``slots[i]`` = decoder state continued by row i (default: slot i).  Returns a list of int16 arrays
        (F_i*160,).  Rows are dispatched longest first (one workgroup per row, row order = dispatch order), so
        when there are more rows than CUs the short ones fill in behind the long ones."""
        n = len(rows)
        if n == 0:
            return []
        counts = np.array([int(np.shape(r)[0]) for r in rows], dtype=np.int32)
        slot_arr = np.arange(n, dtype=np.int32) if slots is None else np.asarray(slots, dtype=np.int32)
        if slot_arr.shape != (n,):
            raise ValueError("slots must have one entry per row")
        fmax = int(counts.max())
        if fmax == 0:
            return [np.empty(0, dtype=np.int16) for _ in rows]
        order = np.argsort(-counts, kind="stable") if longest_first else np.arange(n)
        feats = np.zeros((n, fmax, NB_FEATURES), dtype=np.float32)
        for k, i in enumerate(order):
            if counts[i]:
                r = np.asarray(rows[i], dtype=np.float32)
                if r.ndim != 2 or r.shape[1] < NB_FEATURES:
                    raise ValueError("each row must be (F_i, >=20)")
                feats[k, :counts[i]] = r[:, :NB_FEATURES]
        c_sorted = np.ascontiguousarray(counts[order])
        s_sorted = np.ascontiguousarray(slot_arr[order])
        pcm = np.empty((n, fmax * FRAME_SIZE), dtype=np.int16)
        _lib.check(self._L.dss_lpcnet_batch_synthesize_ragged(self._h, feats.ctypes.data, s_sorted.ctypes.data,
                                                              c_sorted.ctypes.data, n, fmax, NB_FEATURES, pcm.ctypes.data))
        out = [None] * n
        for k, i in enumerate(order):
            out[i] = pcm[k, :counts[i] * FRAME_SIZE].copy()
        return out

    def synthesize_ragged_torch(self, features, counts, slots=None, out=None, stream=None):
        """Device-resident ragged form: features CUDA float32 (n, Fmax, S>=20), ``counts``/``slots`` host int
        sequences.  Returns the int16 CUDA tensor (n, Fmax*160); row i is valid up to counts[i]*160."""
        import torch
        assert features.is_cuda and features.dtype == torch.float32 and features.is_contiguous()
        self._check_device(features)
        n, F, S = features.shape
        c = np.ascontiguousarray(counts, dtype=np.int32)
        sl = None if slots is None else np.ascontiguousarray(slots, dtype=np.int32)
        if c.shape != (n,) or (sl is not None and sl.shape != (n,)):
            raise ValueError("counts/slots must have one entry per row")
        if out is None:
            out = torch.empty((n, F * FRAME_SIZE), dtype=torch.int16, device=features.device)
        s = torch.cuda.current_stream(features.device).cuda_stream if stream is None else stream
        _lib.check(self._L.dss_lpcnet_batch_synthesize_ragged_dev(
            self._h, features.data_ptr(), None if sl is None else sl.ctypes.data, c.ctypes.data, n, F, S, out.data_ptr(), s))
        return out

    # ---- test / measurement taps ------------------------------------------------------------------------
    def enable_trace(self, on=True):
        _lib.check(self._L.dss_lpcnet_batch_enable_trace(self._h, int(on)))

    def force_excitation(self, exc, n_frames: int):
        """Teacher forcing (needs enable_trace): exc (n_utts, n_frames*160) uint8, or None to go back to sampling."""
        if exc is None:
            _lib.check(self._L.dss_lpcnet_batch_force_excitation(self._h, None, 0, 0))
            return
        e = np.ascontiguousarray(exc, dtype=np.uint8)
        assert e.ndim == 2 and e.shape[1] == n_frames * FRAME_SIZE
        _lib.check(self._L.dss_lpcnet_batch_force_excitation(self._h, e.ctypes.data, e.shape[0], int(n_frames)))

    def set_multi(self, utterances_per_workgroup: int = 0):
        """Utterances per workgroup: 0 = automatic (two once the call has more utterances than the chip has CUs), 1 or -1 =
        always one (latency kernel), 2 = always two (packed-pair kernel); dss_lpcnet_batch_set_multi."""
        _lib.check(self._L.dss_lpcnet_batch_set_multi(self._h, int(utterances_per_workgroup)))

    def enable_timing(self, on=True):
        _lib.check(self._L.dss_lpcnet_batch_enable_timing(self._h, int(on)))

    def kernel_ms(self, which=0) -> float:
        return float(self._L.dss_lpcnet_batch_kernel_ms(self._h, which))

    def tap(self, utt: int, which: int, n_frames: int) -> np.ndarray:
        width = {0: 1152, 1: 48, 2: 16, 3: FRAME_SIZE, 4: FRAME_SIZE, 5: FRAME_SIZE * 256}[which]
        out = np.empty((n_frames, width), dtype=np.float32)
        _lib.check(self._L.dss_lpcnet_batch_tap(self._h, utt, which, out.ctypes.data, out.size))
        return out
