"""Whole-segment decoding and vocoding OFF the tick path of the many-stream gated mode.

The reference has one stream and blocks it while a completed speech segment goes through the bidirectional decoder and the
vocoder (local/units.py:499-508, 531-538).  With 128 streams on one GPU the tick -- extractor, detector, gate for ALL
streams every 40 ms -- must not wait for the 3.5 s of audio one of them has just finished.  ``SegmentSynthesisQueue`` takes
the segments that closed on a tick, copies them out of the gate's rings into a pool (one launch, on the tick's stream) and
runs them on side streams:

    one ragged decoder call  (dss_dec_forward_rows_dev: every segment whole, from a fresh state, backward pass from its own end)
    one ragged vocoder call  (a LANE of the streams' LPCNetBatch: row i continues the vocoder state of ITS stream, units.py:524)
    one asynchronous copy of the PCM into page-locked host memory, one event

per job, issued by a WORKER THREAD of the queue (the HIP calls of a job, ~0.2 ms of host time on a cold GPU, and the copy of
finished PCM out of the page-locked buffers are then no part of any tick; the thread sleeps on a condition variable while
nothing is open and looks at its lanes' events every 0.2 ms while something is).  A job takes what is waiting when a lane is free -- at most one segment per stream, a stream's segments strictly in
closing order (its next segment waits until the previous one's job has finished: the vocoder state carries over) -- so
segments that closed on different streams are synthesised side by side and nothing is ever reordered within a stream.
``poll()`` returns the segments finished since the last call; it never waits.  ``threaded=False`` keeps everything on the
caller's thread (jobs are then launched and retired inside ``poll()`` / ``drain()``): the blocking mode uses it.

Lanes: each has its own HIP stream, decoder scratch, vocoder scratch and result buffer.  ROCm maps streams onto
``GPU_MAX_HW_QUEUES`` hardware queues (4 by default) and two streams that share one run their kernels one after the other:
a lane sharing the TICK's queue would put 140 ms of vocoder in front of a tick.  So the number of lanes defaults to
``GPU_MAX_HW_QUEUES - 1`` (three without the variable; a host that wants more sets it before the HIP runtime starts --
bench.py and tools/gated_leg.py use 8), at most 7.
"""
from __future__ import annotations

import collections
import ctypes as C
import threading
import time
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .lpcnet import FRAME_SIZE, LPCNetBatch


def default_lanes() -> int:
    """One hardware queue stays the tick's own: GPU_MAX_HW_QUEUES - 1 lanes (ROCm's default of 4 queues: three), at most 7."""
    import os
    try:
        q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        q = 4
    return max(1, min(7, q - 1))


def select_job(pending, busy, max_rows):
    """Which of the waiting segments (objects with a ``stream`` attribute, in closing order) start together: at most `max_rows`, at
    most one per stream, none of a stream that has a job in flight (``busy[stream]``) -- and never a segment whose stream has an
    EARLIER segment still waiting, whatever the reason that one was passed over: a stream's vocoder state carries from segment
    to segment (units.py:524), so its segments run strictly in closing order.  Returns (job, the rest in unchanged order)."""
    job, keep, seen = [], collections.deque(), set()
    for sg in pending:
        if len(job) < max_rows and not busy[sg.stream] and sg.stream not in seen:
            job.append(sg)
        else:
            keep.append(sg)
        seen.add(sg.stream)
    return job, keep


class _Segment:
    __slots__ = ("stream", "length", "row", "tag", "ready", "t_close")

    def __init__(self, stream, length, row, tag, ready, t_close):
        self.stream, self.length, self.row, self.tag, self.ready, self.t_close = stream, length, row, tag, ready, t_close


class _Ready:
    """The event behind one tick's collect launch, shared by that tick's segments."""
    __slots__ = ("event", "refs")

    def __init__(self, event):
        self.event, self.refs = event, 0


class _Lane:
    def __init__(self, L, vocoder: LPCNetBatch, decoder_factory, rows: int, seg_cap: int, n_out: int):
        self.L = L
        self.stream = L.dss_stream_create()
        self.done = L.dss_event_create()
        if not self.stream or not self.done:
            raise _lib.DssError(L.dss_last_error().decode())
        self.torch_stream = torch.cuda.ExternalStream(self.stream)
        self.voc = vocoder.create_lane(rows, seg_cap)
        self.dec = decoder_factory(rows, seg_cap) if decoder_factory is not None else None
        self.feats = torch.zeros(rows * seg_cap * n_out, dtype=torch.float32, device="cuda")      # a job's (rows, fmax, n_out), packed
        self.pcm = torch.empty(rows * seg_cap * FRAME_SIZE, dtype=torch.int16, device="cuda")    # a job's (rows, fmax * 160), packed
        self.host_bytes = rows * seg_cap * FRAME_SIZE * 2
        self.host_ptr = L.dss_host_alloc(self.host_bytes, 1)
        if not self.host_ptr:
            raise MemoryError(L.dss_last_error().decode())
        self.host = np.ctypeslib.as_array((C.c_int16 * (rows * seg_cap * FRAME_SIZE)).from_address(self.host_ptr))
        self.job: Optional[List[_Segment]] = None
        self.job_frames = 0

    def close(self):
        L = self.L
        if getattr(self, "stream", None):
            L.dss_stream_synchronize(self.stream)
            self.voc.close()
            self.dec = None
            L.dss_host_free(self.host_ptr)
            L.dss_event_destroy(self.done)
            L.dss_stream_destroy(self.stream)
            self.stream = None


class SegmentSynthesisQueue:
    def __init__(self, gate, vocoder: LPCNetBatch, n_features: int, seg_cap: int, decoder_factory=None, decoder_module=None,
                 n_lanes: Optional[int] = None, rows_per_job: int = 32, pool_rows: Optional[int] = None, n_out: int = 20,
                 threaded: bool = True):
        """gate: the SpeechGateGPU whose completed segments are taken; vocoder: the LPCNetBatch with one slot per stream.
        decoder_factory(rows, frames) -> BiLstmDecoderGPU (one per lane: each owns its layer buffers), or None: then
        decoder_module (any torch module with the reference's call signature) runs row by row on the lane's stream."""
        self._L = _lib.require_gpu()
        self.gate, self.S, self.C = gate, gate.S, int(n_features)
        self.cap, self.R, self.n_out = int(seg_cap), int(rows_per_job), int(n_out)
        self.module = decoder_module
        if decoder_factory is None and decoder_module is None:
            raise ValueError("a decoder kernel factory or a decoder module is needed")
        if n_lanes is None:
            n_lanes = default_lanes()
        self.lanes = [_Lane(self._L, vocoder, decoder_factory, self.R, self.cap, self.n_out) for _ in range(int(n_lanes))]
        rows = int(pool_rows or max(2 * self.S, 4 * self.R))
        self.pool = torch.zeros((rows, self.cap, self.C), dtype=torch.float32, device="cuda")
        self._free_rows = list(range(rows - 1, -1, -1))
        self._free_events: list = []
        self._all_events: list = []
        self.pending: collections.deque = collections.deque()
        self.busy = np.zeros(self.S, dtype=bool)              # streams with a job in flight
        self.finished: list = []
        self.latencies_ms: list = []                          # segment closed (submit) -> PCM seen on the host (poll)
        self.jobs_launched = 0
        self.segments_done = 0
        self._open = 0                                        # segments submitted and not yet retired
        self.device = vocoder.device
        self._warm_up()
        # shared between the caller's thread (submit / poll / drain) and the worker: pending, the free lists, finished, counters
        self._cv = threading.Condition(threading.RLock())
        self._stop = False
        self._error = None
        self._thread = None
        if threaded:
            self._thread = threading.Thread(target=self._worker, name="dss-segment-queue", daemon=True)
            self._thread.start()

    def _warm_up(self):
        """The first launch of the sample-rate kernels in a process loads their code object (milliseconds): pay that here, on a
        scratch decoder, not on the tick that closes the first segment."""
        scratch = LPCNetBatch(1, 1)
        lane = scratch.create_lane(1, 2)
        lane.synthesize_ragged_torch(torch.zeros((1, 2, 20), dtype=torch.float32, device="cuda"), [2], slots=[0], stream=self.lanes[0].stream)
        _lib.check(self._L.dss_stream_synchronize(self.lanes[0].stream))
        lane.close()
        scratch.close()

    # ---- tick side ------------------------------------------------------------------------------------------
    def submit(self, streams, events, lengths, tags, tick_stream=None):
        """Take segments (streams[i], events[i]) of the gate's LAST push (lengths[i] frames; tags[i] is handed back with the
        PCM).  One collect launch + one event on the tick's stream; nothing waits."""
        n = len(streams)
        if n == 0:
            return
        L = self._L
        ts = torch.cuda.current_stream().cuda_stream if tick_stream is None else tick_stream
        for ln in lengths:
            if ln > self.cap:
                raise ValueError(f"segment of {ln} frames exceeds max_segment_frames={self.cap}")
        self._raise_worker_error()
        with self._cv:
            while len(self._free_rows) < n:                   # pool exhausted: wait for a job (never in a paced run)
                if self._thread is not None:
                    self._cv.wait(0.001)
                    self._raise_worker_error()
                else:
                    self._dispatch()
                    if not self._wait_one():
                        raise RuntimeError("segment pool exhausted with no job in flight")
            rows = [self._free_rows.pop() for _ in range(n)]
            ev = self._free_events.pop() if self._free_events else None
        self.gate.collect_torch(streams, events, rows, self.pool, hip_stream=ts)
        if ev is None:
            ev = L.dss_event_create()
            if not ev:
                raise _lib.DssError(L.dss_last_error().decode())
            self._all_events.append(ev)
        _lib.check(L.dss_event_record(ev, ts))
        ready = _Ready(ev)
        now = time.perf_counter()
        with self._cv:
            for s, ln, row, tag in zip(streams, lengths, rows, tags):
                ready.refs += 1
                self.pending.append(_Segment(int(s), int(ln), row, tag, ready, now))
            self._open += n
            self._cv.notify_all()

    def _raise_worker_error(self):
        if self._error is not None:
            e, self._error = self._error, None
            raise RuntimeError("segment synthesis worker failed") from e

    # ---- side streams -----------------------------------------------------------------------------------------
    def _launch(self, lane: _Lane, job: List[_Segment]):
        L = self._L
        n = len(job)
        counts = np.fromiter((sg.length for sg in job), dtype=np.int32, count=n)
        rows = np.fromiter((sg.row for sg in job), dtype=np.int32, count=n)
        slots = np.fromiter((sg.stream for sg in job), dtype=np.int32, count=n)
        fmax = max(1, int(counts.max()))
        _lib.check(L.dss_stream_wait_event(lane.stream, job[-1].ready.event))     # the newest segment's collect covers the older ones
        feats = lane.feats[: self.R * fmax * self.n_out].view(self.R, fmax, self.n_out)
        if lane.dec is not None:
            lane.dec.forward_rows_torch(self.pool, rows, counts, feats, fmax, stream=lane.stream)
        else:                                                 # a decoder of another architecture: the module, row by row
            with torch.cuda.stream(lane.torch_stream), torch.no_grad():
                for k, sg in enumerate(job):
                    if sg.length:
                        x = self.pool[sg.row, : sg.length][None]
                        y, _ = self.module(x, self.module.create_new_initial_state(batch_size=1, device="cuda"))
                        feats[k, : sg.length] = y[0]
        pcm = lane.pcm[: self.R * fmax * FRAME_SIZE].view(self.R, fmax * FRAME_SIZE)
        lane.voc.synthesize_ragged_torch(feats[:n], counts, slots=slots, out=pcm, stream=lane.stream)
        _lib.check(L.dss_memcpy_d2h_async(lane.host_ptr, pcm.data_ptr(), n * fmax * FRAME_SIZE * 2, lane.stream))
        _lib.check(L.dss_event_record(lane.done, lane.stream))
        lane.job, lane.job_frames = job, fmax
        self.jobs_launched += 1
        for sg in job:
            sg.ready.refs -= 1
            if sg.ready.refs == 0:                            # every segment of that tick has been launched behind a wait on it
                self._free_events.append(sg.ready.event)      # (list.append: atomic; submit pops under the lock)

    def _retire(self, lane: _Lane):
        job, fmax = lane.job, lane.job_frames
        host = lane.host[: len(job) * fmax * FRAME_SIZE].reshape(len(job), fmax * FRAME_SIZE)
        now = time.perf_counter()
        done = [(sg.stream, sg.tag, host[k, : sg.length * FRAME_SIZE].copy()) for k, sg in enumerate(job)]
        lane.job = None
        with self._cv:
            self.finished += done
            for sg in job:
                self.latencies_ms.append((now - sg.t_close) * 1e3)
                self.busy[sg.stream] = False
                self._free_rows.append(sg.row)
            self.segments_done += len(job)
            self._open -= len(job)
            self._cv.notify_all()

    def _wait_one(self) -> bool:
        for lane in self.lanes:
            if lane.job is not None:
                _lib.check(self._L.dss_event_synchronize(lane.done))
                self._retire(lane)
                return True
        return False

    def _take_job(self):
        """The next job out of `pending` (call with the lock held); marks its streams busy."""
        job, keep = select_job(self.pending, self.busy, self.R)
        if job:
            self.pending = keep
            for sg in job:
                self.busy[sg.stream] = True
        return job

    def _dispatch(self) -> bool:
        did = False
        for lane in self.lanes:
            if lane.job is not None:
                continue
            with self._cv:
                job = self._take_job() if self.pending else []
            if not job:
                break
            self._launch(lane, job)
            did = True
        return did

    def _step(self) -> bool:
        """Retire what has finished, start what can start (worker thread, or the caller's in the unthreaded form)."""
        did = False
        for lane in self.lanes:
            if lane.job is not None and _lib.check(self._L.dss_event_query(lane.done)) == 1:
                self._retire(lane)
                did = True
        return self._dispatch() or did

    def _worker(self):
        try:
            _lib.check(self._L.dss_set_device(self.device))
            torch.cuda.set_device(self.device)
            while True:
                did = self._step()
                with self._cv:
                    if self._stop:
                        return
                    if self._open == 0:
                        self._cv.wait()                       # nothing open: sleep until submit() or close()
                    elif not did:
                        self._cv.wait(0.0002)                 # jobs in flight: look at their events again in 0.2 ms (or on submit)
        except BaseException as e:                            # surfaces on the caller's next submit / poll / drain
            self._error = e
            with self._cv:
                self._cv.notify_all()

    def poll(self):
        """Segments finished since the last call, as (stream, tag, pcm int16 host array), a stream's own in closing order.
        Never blocks.  (Unthreaded form: also retires finished jobs and starts waiting segments on free lanes.)"""
        self._raise_worker_error()
        if self._thread is None:
            self._step()
        with self._cv:
            out, self.finished = self.finished, []
        return out

    @property
    def in_flight(self) -> int:
        return self._open

    def drain(self):
        """Wait for everything submitted so far; returns what poll() would have returned over that time."""
        out = self.poll()
        while self._open:
            if self._thread is not None:
                with self._cv:
                    if self._open and not self.finished and self._error is None:
                        self._cv.wait(0.001)
            elif not self._wait_one():
                self._dispatch()
            out += self.poll()
        return out

    def close(self):
        th = getattr(self, "_thread", None)
        if th is not None:
            with self._cv:
                self._stop = True
                self._cv.notify_all()
            th.join(timeout=5.0)
            self._thread = None
        for lane in getattr(self, "lanes", []):
            lane.close()
        self.lanes = []
        for ev in getattr(self, "_all_events", []):
            self._L.dss_event_destroy(ev)
        self._all_events, self._free_events = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
