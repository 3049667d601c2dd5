"""The speech-gate oracle (oracle/speech_gate_oracle.py) against (a) what the reference's own classes returned
(tests/golden/gate.npz: VoiceActivityDetectionSmoothing + SpeechSegmentHistory of /root/reference/local/common.py:106-215, run by
oracle/make_golden.py) and (b) known answers derived by hand from their rules."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def test_gate_oracle_known_answers():
    from speech_gate_oracle import SpeechGateOracle
    # smoothing alone (context 0 in the history => every closed run is emitted at its first non-speech frame)
    g = SpeechGateOracle(nb_features=1, buffer_size=64, context=0, smoothing_context=5)
    data = np.arange(40, dtype=np.float32).reshape(40, 1)
    labels = np.array([0] * 3 + [1] * 12 + [0] * 25)
    segs, n_speech = g.push(data, labels)
    # the smoothed label is on while >= 7 of the last 11 raw labels are speech: inserts 9..18 (10 frames); frames
    # leave the window 10 inserts late, so those inserts carry frames -1..8 -> zeros (still the initial window) for the
    # first one, then frames 0..8
    assert n_speech == 10 and len(segs) == 1
    assert segs[0][:, 0].tolist() == [0.0] + [float(v) for v in range(0, 9)]
    # history with context: 6 speech frames + 3 frames on both sides, closed by the 3rd trailing non-speech frame
    h = SpeechGateOracle(nb_features=1, buffer_size=50, context=3, smoothing_context=0, threshold=0.6)
    segs, _ = h.push(np.arange(30, dtype=np.float32).reshape(30, 1), np.array([0] * 10 + [1] * 6 + [0] * 14))
    assert len(segs) == 1 and segs[0][:, 0].tolist() == list(range(7, 19))


def test_smoothing_and_history_rules_one_at_a_time():
    from speech_gate_oracle import SpeechGateOracle
    # smoothing: frames leave 2*ctx inserts late, the label turns on once >= 60 % (7 of 11) of the window is speech
    g = SpeechGateOracle(nb_features=2, buffer_size=64, context=0, smoothing_context=5)
    assert g.W == 11
    data = np.arange(40, dtype=np.float32).reshape(20, 2)
    segs, n_speech = g.push(data, np.array([0] * 3 + [1] * 12 + [0] * 5))
    assert n_speech == 10 and len(segs) == 1 and segs[0].shape == (10, 2)
    assert np.array_equal(segs[0][1:], data[:9]) and not segs[0][0].any()
    # history: nothing is emitted without speech; ring wrap-around keeps frame order
    h = SpeechGateOracle(nb_features=1, buffer_size=16, context=2, smoothing_context=0)
    ramp = np.arange(30, dtype=np.float32).reshape(30, 1)
    assert h.push(ramp[:12], np.zeros(12, dtype=bool)) == ([], 0)
    segs, _ = h.push(ramp[12:24], np.array([1] * 6 + [0] * 6, dtype=bool))
    assert segs[0][:, 0].tolist() == list(range(10, 20))
    assert h.frames_seen == 24


def replay_gate_fixture(g, ci, push):
    """Feed case `ci` of tests/golden/gate.npz, push by push, to push(frames, labels) -> (segments, speech frames) and compare
    with what the reference classes returned.  Shared with tests/test_gpu_gate.py."""
    C, N, ctx, sm = (int(v) for v in g[f"case{ci}_params"])
    sizes, frames, labels = g[f"case{ci}_sizes"], g[f"case{ci}_frames"].astype(np.float64), g[f"case{ci}_labels"]
    seg_push, seg_len, seg_rows = g[f"case{ci}_seg_push"], g[f"case{ci}_seg_len"], g[f"case{ci}_seg_rows"]
    want_speech = g[f"case{ci}_speech_per_push"]
    assert frames.shape == (int(sizes.sum()), C) and seg_rows.shape == (int(seg_len.sum()), C) and seg_rows.dtype == np.float32
    pos, seg_i, row0 = 0, 0, 0
    for k, W in enumerate(sizes):
        segs, n_speech = push(frames[pos:pos + W], labels[pos:pos + W].astype(np.int64))
        pos += int(W)
        assert n_speech == int(want_speech[k]), (ci, k)
        for seg in segs:
            assert seg_i < len(seg_push) and int(seg_push[seg_i]) == k, (ci, k, "a segment the reference did not emit here")
            L = int(seg_len[seg_i])
            assert seg.dtype == np.float32 and seg.shape == (L, C) and np.array_equal(seg, seg_rows[row0:row0 + L]), (ci, k)
            seg_i, row0 = seg_i + 1, row0 + L
        assert seg_i == int(np.searchsorted(seg_push, k, side="right")), (ci, k, "a segment of the reference is missing")
    assert seg_i == len(seg_push)
    return (C, N, ctx, sm), len(seg_push)


def test_gate_oracle_against_the_reference_classes(golden):
    from speech_gate_oracle import SpeechGateOracle
    g = golden("gate.npz")
    total = 0
    for ci in range(int(g["n_cases"][0])):
        C, N, ctx, sm = (int(v) for v in g[f"case{ci}_params"])
        gate = SpeechGateOracle(C, N, ctx, sm)
        total += replay_gate_fixture(g, ci, gate.push)[1]
    assert total > 100
    # the cases the fixture is meant to hold: a ring that wraps (segment count x length far beyond N), context 0, and a run
    # longer than its ring (the emitted "segment" then has fewer rows than the run: the classes' modulo arithmetic)
    assert int(g["case1_seg_len"].sum()) > 5 * 37 and int(g["case2_params"][2]) == 0
    assert int(g["case3_seg_len"].max()) < 23
