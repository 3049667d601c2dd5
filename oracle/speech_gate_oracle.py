"""oracle/speech_gate_oracle.py -- CPU restatement of the reference's speech-segment gating (TEST INFRASTRUCTURE ONLY:
imported by tests/ alone; the product never touches oracle/).

One object = one stream.  It restates, frame by frame and in plain Python/numpy, what FilterSpeechSegments.process
(reference local/units.py:432-447) does with the two ring-buffer classes of reference local/common.py:

  smoothing  VoiceActivityDetectionSmoothing.__init__/insert   common.py:113-147
             window of 2*ctx+1 raw labels; write pointer starts at 2*ctx, read pointer at 0; a frame is speech when
             count_nonzero(window) / len(window) >= threshold (float64); the frame leaving is the one at the read pointer
  history    SpeechSegmentHistory.__init__/insert              common.py:160-215
             ring of N float32 frames; on the context-th non-speech frame after >= 1 speech frame the segment
             [stop - 2*context - speech_count, stop) modulo N is emitted (stop = write pointer, or write pointer - 1 when
             context == 0) and both counters restart

Pinned against the reference's OWN classes: tests/golden/gate.npz holds what VoiceActivityDetectionSmoothing +
SpeechSegmentHistory of /root/reference/local/common.py returned for seeded label runs in five configurations (rings that
wrap, context 0, runs longer than the ring; oracle/make_golden.py gen_common imports the file as it lies), and
tests/test_oracle_gate.py replays them through this restatement bit for bit; the hand-derived known answers of rounds 1-4
stay beside them.
"""
import numpy as np


class SpeechGateOracle:
    def __init__(self, nb_features, buffer_size, context=0, smoothing_context=5, threshold=0.6):
        self.C, self.N, self.ctx, self.thr = nb_features, buffer_size, context, threshold
        self.W = 2 * smoothing_context + 1
        self.win_frames = np.zeros((self.W, nb_features), np.float32)
        self.win_labels = [False] * self.W
        self.w, self.r = 2 * smoothing_context, 0
        self.ring = np.zeros((buffer_size, nb_features), np.float32)
        self.hw = 0
        self.speech = 0
        self.after = 0
        self.frames_seen = 0

    def push(self, frames, labels):
        """frames (n, C) any float, labels (n,) -> (list of completed segments (L, C) float32, speech frames in this push)"""
        segments, n_speech = [], 0
        for x, raw in zip(np.asarray(frames), np.asarray(labels)):
            # smoothing (common.py:130-147)
            self.win_labels[self.w] = bool(raw)
            self.win_frames[self.w] = x
            is_speech = (sum(self.win_labels) / self.W) >= self.thr
            out = self.win_frames[self.r].copy()
            self.w = (self.w + 1) % self.W
            self.r = (self.r + 1) % self.W
            # history (common.py:193-214)
            self.ring[self.hw] = out
            self.hw = (self.hw + 1) % self.N
            if is_speech:
                self.speech += 1
                n_speech += 1
            elif self.speech > 0:
                self.after += 1
                if self.after >= self.ctx:
                    stop = self.hw if self.ctx > 0 else (self.hw - 1) % self.N
                    start = (stop - 2 * self.ctx - self.speech) % self.N
                    rows = []
                    p = start
                    while p != stop:                      # common.py:172-181 (_get_positions)
                        rows.append(p)
                        p = (p + 1) % self.N
                    segments.append(self.ring[rows].copy() if rows else np.zeros((0, self.C), np.float32))
                    self.speech = self.after = 0
        self.frames_seen += len(labels)
        return segments, n_speech
