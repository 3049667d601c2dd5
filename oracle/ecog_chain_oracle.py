"""oracle/ecog_chain_oracle.py -- CPU restatement of the reference's pre/post transform chain around the HGA extractor
(TEST INFRASTRUCTURE ONLY: imported by tests/ alone; the product never touches oracle/).

decode_online.py:65-97 builds, in this order, in front of the filters
    SelectElectrodesFromBothGrids      local/common.py:16-32    data[:, grid_mapping]           (129 raw columns -> chan1..chan128)
    CommonAverageReferencing           local/common.py:308-345  per grid: minus the mean over the grid's non-excluded channels
    SelectElectrodesOverSpeechAreas    local/common.py:35-58    data[:, speech_grid_mapping]    (-> 64 channels)
and behind the log power
    ZScoreNormalization                local/common.py:367-376  (data - means) / stds
Pinned against the reference's OWN classes: tests/golden/ecog_chain.npz holds every stage of that chain as
/root/reference/local/common.py computed it on a seeded 129-column packet (oracle/make_golden.py gen_common imports the file
as it lies), and tests/test_cpu_host.py compares this restatement with it bit for bit, beside the closed-form checks of
rounds 1-4.
The objects expose the attribute names of the reference classes (grid_mapping, selection_masks_application,
selection_masks_computation, speech_grid_mapping): the product recognises decode_online.py's chain through them.
The channel tables are the product's data module (dss_amd/electrodes.py), i.e. the constants of common.py:21-28,42-48.
"""
import numpy as np


class ReorderBothGrids:                                 # common.py:16-32
    def __init__(self, grid_columns):
        self.grid_mapping = list(grid_columns)

    def __call__(self, data):
        return data[:, self.grid_mapping]


class SelectSpeechArea:                                 # common.py:35-58
    def __init__(self, speech_area_zero_based, bad_channels_one_based):
        kept = [c for c in speech_area_zero_based if (c + 1) not in set(bad_channels_one_based)]
        self.speech_grid_mapping = np.array(sorted(kept))

    def __call__(self, data):
        return data[:, self.speech_grid_mapping]


class GridCommonAverage:                                # common.py:308-345
    def __init__(self, exclude_channels, grids, layout):
        layout = np.asarray(layout)
        self.selection_masks_application = [np.isin(layout, g) for g in grids]
        self.selection_masks_computation = []
        for g, applied in zip(grids, self.selection_masks_application):
            used = applied.copy()
            for ch in exclude_channels:
                if ch in g:
                    used[np.argmax(layout == ch)] = False           # common.py:331-333
            self.selection_masks_computation.append(used)

    def __call__(self, data):
        out = data.copy()
        for used, applied in zip(self.selection_masks_computation, self.selection_masks_application):
            # np.mean over a fancy-indexed (T, k) copy along axis 1 (common.py:340): the copy is Fortran-ordered, so
            # numpy adds one column at a time -- a sequential sum in ascending channel order, then one division
            mean = np.mean(data[:, used], axis=1).reshape(-1, 1)
            out[:, applied] = out[:, applied] - mean                 # common.py:341-343 (tile + subtract)
        return out


class ZScore:                                           # common.py:367-376
    def __init__(self, channel_means, channel_stds):          # attribute names as in the reference (common.py:371-372)
        self.channel_means, self.channel_stds = channel_means, channel_stds

    def __call__(self, data):
        return (data - self.channel_means) / self.channel_stds


def reference_chain(bad_channels=(19, 38, 48, 52)):
    """The three pre-transform objects exactly as decode_online.py:65-85 configures them."""
    import os, sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "delayed-speech-synthesis_amd")
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    from dss_amd.electrodes import GRID_COLUMNS, SPEECH_AREA_CHANNELS
    speech_grid = np.flip(np.arange(64, dtype=np.int16).reshape((8, 8)) + 1, axis=0)       # decode_online.py:67-70
    motor_grid = np.flip(np.arange(64, dtype=np.int16).reshape((8, 8)) + 65, axis=0)
    return (ReorderBothGrids(GRID_COLUMNS),
            GridCommonAverage(list(bad_channels), [speech_grid, motor_grid], np.arange(128) + 1),
            SelectSpeechArea(SPEECH_AREA_CHANNELS, bad_channels))
