"""The speech-gate oracle (oracle/speech_gate_oracle.py) against known answers derived by hand from the rules of
reference local/common.py:106-215 (the reference has no test or fixture for these classes)."""
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
