"""Feature transforms and segment bookkeeping around the hot path: channel
selection, per-grid common average referencing, z-scoring, VAD label smoothing and the speech-segment ring
buffer.  Behaviour follows the reference's local/common.py (classes of the same names); the electrode tables
are the subject-specific constants of that file (common.py:21-28, 42-49) and are data, not logic."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

# column of the 128-channel amplifier packet that holds grid channel 1, 2, ..., 128   (common.py:21-28)
_GRID_COLUMNS = (
    125, 123, 121, 119, 122, 111, 118, 124, 120, 126, 127, 116, 114, 113, 115, 117, 98, 97, 96, 104, 100, 102, 101, 99,
    105, 112, 107, 106, 108, 103, 109, 110, 17, 21, 9, 28, 26, 31, 13, 27, 25, 22, 30, 11, 29, 23, 19, 15, 1, 2, 4, 0,
    24, 12, 14, 7, 5, 18, 6, 10, 3, 8, 20, 16, 50, 33, 44, 51, 63, 40, 38, 46, 42, 48, 56, 37, 35, 41, 47, 58, 61, 60,
    59, 43, 49, 45, 54, 62, 32, 53, 55, 52, 57, 39, 34, 36, 85, 84, 83, 87, 80, 86, 90, 78, 75, 92, 76, 88, 82, 94, 70,
    74, 69, 66, 79, 71, 73, 77, 68, 67, 64, 65, 95, 93, 81, 72, 91, 89)
# zero-based grid channels over speech areas before bad-channel removal   (common.py:42-46)
_SPEECH_AREA = (1, 2, 3, 0, 4, 11, 5, 6, 7, 10, 12, 9, 19, 8, 15, 20, 13, 14, 17, 22, 18, 21, 29, 16, 23, 28, 35, 36, 27,
                25, 26, 55, 45, 46, 44, 24, 37, 40, 33, 34, 32, 51, 47, 39, 31, 54, 53, 30, 48, 38, 43, 41, 52, 61, 59, 62,
                49, 66, 60, 63, 58, 50, 42, 56, 67, 57, 81, 68)
_BAD_CHANNELS = (19, 38, 48, 52)          # one-based   (common.py:48)


class SelectElectrodesFromBothGrids:
    """Reorder the amplifier columns to chan1, chan2, ..., chan128."""

    def __init__(self):
        self.grid_mapping = list(_GRID_COLUMNS)

    def __len__(self):
        return len(self.grid_mapping)

    def __call__(self, data):
        return data[:, self.grid_mapping]


class SelectElectrodesOverSpeechAreas:
    """Keep the electrodes over speech areas (64 of them once the four bad channels are dropped)."""

    def __init__(self):
        one_based = np.asarray(_SPEECH_AREA) + 1
        kept = np.array([c for c in one_based if c not in _BAD_CHANNELS])
        self.speech_grid_mapping = np.array(sorted(kept - 1))

    def __len__(self):
        return len(self.speech_grid_mapping)

    def __call__(self, data):
        return data[:, self.speech_grid_mapping]

    def __repr__(self):
        return f"Channels: {', '.join(map(str, self.speech_grid_mapping + 1))}"


class CommonAverageReferencing:
    """Per electrode grid, subtract from every channel of the grid the mean over the grid's non-excluded
    channels at each time point.  data: (T, E)."""

    def __init__(self, exclude_channels: List[int], grids: List[np.ndarray], layout: np.ndarray):
        self.grids, self.layout = grids, np.asarray(layout)
        self.selection_masks_application = [np.isin(self.layout, grid) for grid in grids]
        self.selection_masks_computation = []
        for grid, applied in zip(grids, self.selection_masks_application):
            used = applied.copy()
            for ch in exclude_channels:
                if ch in grid:
                    used[np.argmax(self.layout == ch)] = False
            self.selection_masks_computation.append(used)

    def __call__(self, data: np.ndarray) -> np.ndarray:
        out = data.copy()
        for used, applied in zip(self.selection_masks_computation, self.selection_masks_application):
            mean = np.mean(data[:, used], axis=1).reshape((-1, 1))
            out[:, applied] = out[:, applied] - np.tile(mean, reps=(1, np.count_nonzero(applied)))
        return out


class ZScoreNormalization:
    def __init__(self, channel_means: np.ndarray, channel_stds: np.ndarray):
        self.channel_means, self.channel_stds = channel_means, channel_stds

    def __call__(self, data):
        return (data - self.channel_means) / self.channel_stds


class VoiceActivityDetectionSmoothing:
    """Majority smoothing of VAD labels over a window of 2*context+1 frames; frames leave `context` frames late
    so that data and smoothed labels stay aligned (reference common.py:106-153)."""

    def __init__(self, nb_features: int, context_frames: int, proportion_threshold: float = 0.6, shift: float = 0.01):
        self.frameshift = shift
        self.nb_features = nb_features
        self.vad_context_frames = context_frames
        self.vad_proportion_threshold = proportion_threshold
        self.buffer_size = 2 * context_frames + 1
        self.buffer = np.zeros((self.buffer_size, nb_features), dtype=np.float32)
        self.labels = np.zeros(self.buffer_size, dtype=bool)
        self.write_pointer = 2 * context_frames
        self.read_pointer = 0

    def insert(self, data: np.ndarray, speech_labels: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        n = len(speech_labels)
        out_labels = np.zeros(n, dtype=bool)
        out_data = np.zeros((n, self.nb_features), dtype=np.float32)
        for i in range(n):
            self.labels[self.write_pointer] = speech_labels[i]
            self.buffer[self.write_pointer, :] = data[i]
            out_labels[i] = (np.count_nonzero(self.labels) / self.buffer_size) >= self.vad_proportion_threshold
            out_data[i, :] = self.buffer[self.read_pointer, :]
            self.write_pointer = (self.write_pointer + 1) % self.buffer_size
            self.read_pointer = (self.read_pointer + 1) % self.buffer_size
        return out_data, out_labels


class SpeechSegmentHistory:
    """Ring buffer of frames; when `context` non-speech frames have followed a run of speech frames the whole
    segment (speech plus `context` frames on both sides) is returned (reference common.py:156-215)."""

    def __init__(self, nb_features: int, buffer_size: int, context: int = 0):
        self.buffer = np.zeros((buffer_size, nb_features), dtype=np.float32)
        self.write_pointer = 0
        self.context = context
        self.speech_frame_counter = 0
        self.future_frame_counter = 0

    def insert(self, data: np.ndarray, speech_labels: np.ndarray) -> List[np.ndarray]:
        size = len(self.buffer)
        segments = []
        for frame, label in zip(data, speech_labels):
            self.buffer[self.write_pointer, :] = frame
            self.write_pointer = (self.write_pointer + 1) % size
            if label:
                self.speech_frame_counter += 1
            elif self.speech_frame_counter > 0:
                self.future_frame_counter += 1
                if self.future_frame_counter >= self.context:
                    stop = self.write_pointer if self.context > 0 else (self.write_pointer - 1) % size
                    start = (stop - 2 * self.context - self.speech_frame_counter) % size
                    count = (stop - start) % size
                    positions = (start + np.arange(count)) % size
                    segments.append(self.buffer[positions])
                    self.speech_frame_counter = 0
                    self.future_frame_counter = 0
        return segments
