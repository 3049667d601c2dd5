"""Speech-segment gate for many streams on MI355X: Python host side of Part 4 of include/dss_hip.h.

``SpeechGateGPU`` holds, per stream, the two ring buffers the reference chains behind its neural VAD in
``FilterSpeechSegments.process`` (local/units.py:432-447): ``VoiceActivityDetectionSmoothing`` and
``SpeechSegmentHistory`` (local/common.py:106-215).  The VAD network stays a PyTorch module; its argmax decisions
and the z-scored high-gamma frames go in, completed speech segments (float32, exactly the rows the numpy classes
would return) come out.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import _lib


class SpeechGateGPU:
    def __init__(self, n_streams: int, nb_features: int, buffer_size: int, context_frames: int = 0,
                 smoothing_context: int = 5, proportion_threshold: float = 0.6, max_frames: int = 8):
        L = _lib.require_gpu()
        self._L = L
        self.S, self.C, self.N = int(n_streams), int(nb_features), int(buffer_size)
        self.max_frames = int(max_frames)
        self._h = L.dss_gate_create(self.S, self.C, int(smoothing_context), float(proportion_threshold), self.N,
                                    int(context_frames), self.max_frames)
        if not self._h:
            raise _lib.DssError(L.dss_last_error().decode())
        self.E = int(L.dss_gate_max_events(self._h))
        self._events = np.zeros((self.S, 2 + self.E), dtype=np.int32)

    def close(self):
        if getattr(self, "_h", None):
            self._L.dss_gate_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self, stream: int = -1):
        _lib.check(self._L.dss_gate_reset(self._h, int(stream)))

    def frames_seen(self, stream: int) -> int:
        return _lib.check(self._L.dss_gate_frames_seen(self._h, int(stream)))

    # ---- host buffers ---------------------------------------------------------------------------------------
    def push(self, frames: np.ndarray, labels: np.ndarray) -> Tuple[List[List[np.ndarray]], np.ndarray]:
        """frames (S, W, C) float64, labels (S, W) -> (per stream: list of completed segments (L, C) float32,
        per stream: number of frames of this push the smoothing labelled speech)."""
        f = np.ascontiguousarray(frames, dtype=np.float64)
        lab = np.ascontiguousarray(np.asarray(labels) != 0, dtype=np.int32)
        if f.ndim != 3 or f.shape[0] != self.S or f.shape[2] != self.C or lab.shape != f.shape[:2]:
            raise ValueError(f"expected frames ({self.S}, W, {self.C}) and labels ({self.S}, W)")
        _lib.check(self._L.dss_gate_push(self._h, f.ctypes.data, lab.ctypes.data, f.shape[1], self._events.ctypes.data))
        out = []
        for s in range(self.S):
            segs = []
            for e in range(int(self._events[s, 0])):
                seg = np.empty((int(self._events[s, 2 + e]), self.C), dtype=np.float32)
                _lib.check(self._L.dss_gate_segment(self._h, s, e, seg.ctypes.data, seg.shape[0]))
                segs.append(seg)
            out.append(segs)
        return out, self._events[:, 1].copy()

    # ---- device-resident form -----------------------------------------------------------------------------------
    def push_torch(self, frames, labels) -> np.ndarray:
        """frames CUDA float64 (S, W, C), labels CUDA int32 (S, W).  Returns the host event table (S, 2+E):
        [segments completed, speech-labelled frames, length of segment 0, ...] (synchronises the stream)."""
        import torch
        assert frames.is_cuda and frames.dtype == torch.float64 and frames.is_contiguous()
        assert labels.is_cuda and labels.dtype == torch.int32 and labels.is_contiguous()
        S, W, C = frames.shape
        if S != self.S or C != self.C or tuple(labels.shape) != (S, W):
            raise ValueError(f"expected frames ({self.S}, W, {self.C}) and labels ({self.S}, W)")
        s = torch.cuda.current_stream(frames.device).cuda_stream
        _lib.check(self._L.dss_gate_push_dev(self._h, frames.data_ptr(), labels.data_ptr(), W, self._events.ctypes.data, s))
        return self._events

    def collect_torch(self, streams, events, dst_rows, pool, hip_stream=None):
        """Segments (streams[i], events[i]) of the last push -> rows dst_rows[i] of ``pool`` (CUDA float32 (rows, row_frames, C)) in
        one launch (``dss_gate_collect_dev``), asynchronous on the current stream."""
        import torch
        assert pool.is_cuda and pool.dtype == torch.float32 and pool.is_contiguous() and pool.dim() == 3 and pool.shape[2] == self.C
        st = np.ascontiguousarray(streams, dtype=np.int32)
        ev = np.ascontiguousarray(events, dtype=np.int32)
        dr = np.ascontiguousarray(dst_rows, dtype=np.int32)
        if not (st.shape == ev.shape == dr.shape) or (dr.size and int(dr.max()) >= pool.shape[0]):
            raise ValueError("streams / events / dst_rows must have one entry per segment, rows inside the pool")
        s = torch.cuda.current_stream().cuda_stream if hip_stream is None else hip_stream
        _lib.check(self._L.dss_gate_collect_dev(self._h, int(st.size), st.ctypes.data, ev.ctypes.data, dr.ctypes.data, pool.data_ptr(),
                                                int(pool.shape[1]), s))

    def segment_torch(self, stream: int, event: int = 0):
        """Segment `event` completed by `stream` in the last push, as a CUDA float32 tensor (L, C)."""
        import torch
        length = int(self._events[stream, 2 + event])
        out = torch.empty((length, self.C), dtype=torch.float32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(self._L.dss_gate_segment_dev(self._h, int(stream), int(event), out.data_ptr(), length, s))
        return out
