"""Host logic of the asynchronous segment queue (dss_amd/segment_queue.py) that needs no GPU: which waiting segments may start
together.  The rule it keeps is the reference's: one vocoder per stream whose state carries from segment to segment
(local/units.py:524,531-538), so a stream's segments run strictly in closing order; different streams' segments run side by side."""
import collections
import importlib

import numpy as np


def _module():
    # the module imports torch and the library binding at import time; neither is touched by select_job
    return importlib.import_module("dss_amd.segment_queue")


class Seg:
    def __init__(self, stream, tag):
        self.stream, self.tag = stream, tag

    def __repr__(self):
        return f"{self.stream}:{self.tag}"


def test_select_job_keeps_a_streams_segments_in_closing_order():
    sq = _module()
    busy = np.zeros(6, dtype=bool)
    pending = collections.deque([Seg(0, "a"), Seg(1, "a"), Seg(0, "b"), Seg(2, "a"), Seg(1, "b"), Seg(3, "a")])
    job, keep = sq.select_job(pending, busy, 32)
    assert [repr(s) for s in job] == ["0:a", "1:a", "2:a", "3:a"]              # one per stream, closing order
    assert [repr(s) for s in keep] == ["0:b", "1:b"]
    # a stream with a job in flight: none of its segments starts, the others do
    busy[1] = True
    job, keep = sq.select_job(pending, busy, 32)
    assert [repr(s) for s in job] == ["0:a", "2:a", "3:a"] and [repr(s) for s in keep] == ["1:a", "0:b", "1:b"]
    # a full job: the segment that did not fit blocks the LATER segments of its own stream, not those of other streams
    busy[:] = False
    pending = collections.deque([Seg(0, "a"), Seg(1, "a"), Seg(2, "a"), Seg(2, "b"), Seg(3, "a")])
    job, keep = sq.select_job(pending, busy, 2)
    assert [repr(s) for s in job] == ["0:a", "1:a"] and [repr(s) for s in keep] == ["2:a", "2:b", "3:a"]
    job, keep = sq.select_job(keep, busy, 2)
    assert [repr(s) for s in job] == ["2:a", "3:a"] and [repr(s) for s in keep] == ["2:b"]
    assert sq.select_job(collections.deque(), busy, 4) == ([], collections.deque())


def test_select_job_random_schedules_never_reorder_a_stream():
    sq = _module()
    rng = np.random.default_rng(0)
    for trial in range(50):
        S = int(rng.integers(1, 9))
        pending = collections.deque(Seg(int(rng.integers(S)), k) for k in range(int(rng.integers(0, 40))))
        busy = rng.random(S) < 0.3
        done = {s: [] for s in range(S)}
        in_flight = []
        for step in range(200):
            job, pending = sq.select_job(pending, busy, int(rng.integers(1, 5)))
            assert len({sg.stream for sg in job}) == len(job) and not any(busy[sg.stream] for sg in job)
            for sg in job:
                busy[sg.stream] = True
            in_flight += job
            if in_flight and rng.random() < 0.7:                 # some job finishes
                k = int(rng.integers(len(in_flight)))
                sg = in_flight.pop(k)
                busy[sg.stream] = False
                done[sg.stream].append(sg.tag)
            elif not in_flight:
                busy[:] = False                                   # the streams that were busy at the start have finished too
            if not pending and not in_flight:
                break
        assert not pending and not in_flight, trial
        for s in range(S):
            assert done[s] == sorted(done[s]), (trial, s)         # tags were handed out in closing order


def test_default_lanes_follow_the_hardware_queue_count(monkeypatch):
    sq = _module()
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert sq.default_lanes() == 3                                # ROCm's default of four queues: one stays the tick's
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert sq.default_lanes() == 7
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "64")
    assert sq.default_lanes() == 7
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "1")
    assert sq.default_lanes() == 1
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "many")
    assert sq.default_lanes() == 3
