"""The speech-gate oracle (oracle/speech_gate_oracle.py) against hand-derived known answers and against the host
classes of the drop-in local/common.py (two independent restatements of reference local/common.py:106-215)."""
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


def test_gate_oracle_equals_host_classes_on_random_runs():
    from local.common import SpeechSegmentHistory, VoiceActivityDetectionSmoothing
    from speech_gate_oracle import SpeechGateOracle
    rng = np.random.default_rng(3)
    for C, N, ctx, sm in ((4, 37, 3, 2), (3, 16, 0, 0), (2, 300, 50, 5), (5, 23, 4, 1)):
        orc = SpeechGateOracle(C, N, ctx, sm)
        smo = VoiceActivityDetectionSmoothing(nb_features=C, context_frames=sm)
        hist = SpeechSegmentHistory(nb_features=C, buffer_size=N, context=ctx)
        state, total = 0, 0
        for _ in range(150):
            W = int(rng.integers(1, 8))
            frames = rng.standard_normal((W, C))
            labels = np.zeros(W, dtype=np.int64)
            for i in range(W):
                if rng.random() < 0.08:
                    state = 1 - state
                labels[i] = state
            want_data, want_lab = smo.insert(data=frames, speech_labels=labels)
            want = hist.insert(data=want_data, speech_labels=want_lab)
            got, n_speech = orc.push(frames, labels)
            assert n_speech == np.count_nonzero(want_lab) and len(got) == len(want)
            for a, b in zip(got, want):
                assert a.dtype == np.float32 and np.array_equal(a, b)
            total += len(want)
        assert total > 0
