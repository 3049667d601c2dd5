"""Electrode tables of the reference's recording montage, as DATA, and the fused GPU front-end description built
from them.

The reference's pre-transform chain (decode_online.py:65-85: SelectElectrodesFromBothGrids -> CommonAverageReferencing
-> SelectElectrodesOverSpeechAreas, local/common.py:16-58,308-345) is three numpy gathers and two means per time
sample.  On the GPU it is one kernel ahead of the IIR cascades (csrc/hga_kernels.hip, ``dss_hga_set_frontend``):

    out[:, c] = raw[:, src_col[c]] - mean(raw[:, comp_cols[grid_of[c]]])

This module holds only what that kernel needs: the subject-specific channel tables (constants of
local/common.py:21-28,42-48, data rather than logic) and the index arithmetic that turns them into
``(src_col, grid_of, comp_cols)``.  The numpy classes themselves are the user's own ``local.common``; objects of those
classes are accepted by ``dss_amd.units.HighGammaExtractor`` through the attributes they expose
(``grid_mapping``, ``selection_masks_*``, ``speech_grid_mapping``).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

# column of the amplifier packet that carries grid channel 1, 2, ..., 128 (local/common.py:21-28)
GRID_COLUMNS: Tuple[int, ...] = (
    125, 123, 121, 119, 122, 111, 118, 124, 120, 126, 127, 116, 114, 113, 115, 117, 98, 97, 96, 104, 100, 102, 101, 99,
    105, 112, 107, 106, 108, 103, 109, 110, 17, 21, 9, 28, 26, 31, 13, 27, 25, 22, 30, 11, 29, 23, 19, 15, 1, 2, 4, 0,
    24, 12, 14, 7, 5, 18, 6, 10, 3, 8, 20, 16, 50, 33, 44, 51, 63, 40, 38, 46, 42, 48, 56, 37, 35, 41, 47, 58, 61, 60,
    59, 43, 49, 45, 54, 62, 32, 53, 55, 52, 57, 39, 34, 36, 85, 84, 83, 87, 80, 86, 90, 78, 75, 92, 76, 88, 82, 94, 70,
    74, 69, 66, 79, 71, 73, 77, 68, 67, 64, 65, 95, 93, 81, 72, 91, 89)
# zero-based grid channels over speech areas, bad channels still included (local/common.py:42-46)
SPEECH_AREA_CHANNELS: Tuple[int, ...] = (
    1, 2, 3, 0, 4, 11, 5, 6, 7, 10, 12, 9, 19, 8, 15, 20, 13, 14, 17, 22, 18, 21, 29, 16, 23, 28, 35, 36, 27,
    25, 26, 55, 45, 46, 44, 24, 37, 40, 33, 34, 32, 51, 47, 39, 31, 54, 53, 30, 48, 38, 43, 41, 52, 61, 59, 62,
    49, 66, 60, 63, 58, 50, 42, 56, 67, 57, 81, 68)
BAD_CHANNELS: Tuple[int, ...] = (19, 38, 48, 52)          # one-based (local/common.py:48; decode_online.py:72)
GRIDS_ONE_BASED: Tuple[Tuple[int, int], ...] = ((1, 64), (65, 128))     # speech grid, motor grid (decode_online.py:67-70)


def speech_channels_zero_based(bad_channels: Sequence[int] = BAD_CHANNELS) -> np.ndarray:
    """The 64 zero-based grid channels kept by the speech-area selection, ascending."""
    bad0 = {b - 1 for b in bad_channels}
    return np.array(sorted(c for c in SPEECH_AREA_CHANNELS if c not in bad0), dtype=np.int64)


def reference_frontend(bad_channels: Sequence[int] = BAD_CHANNELS):
    """(src_col, grid_of, comp_lists) of decode_online.py's pre-transform chain for
    ``HgaExtractorGPU.set_frontend``: 64 output channels, two grids, bad channels left out of the means."""
    cols = np.asarray(GRID_COLUMNS, dtype=np.int64)                 # grid channel k (0-based) sits in raw column cols[k]
    keep = speech_channels_zero_based(bad_channels)
    src_col = cols[keep]
    grid_of = np.array([next(g for g, (lo, hi) in enumerate(GRIDS_ONE_BASED) if lo <= k + 1 <= hi) for k in keep])
    bad0 = {b - 1 for b in bad_channels}
    comp_lists: List[np.ndarray] = [cols[[k for k in range(lo - 1, hi) if k not in bad0]] for lo, hi in GRIDS_ONE_BASED]
    return src_col, grid_of, comp_lists


def frontend_from_transforms(select_all, car, select_sub):
    """Same triple from three transform OBJECTS of the user's ``local.common`` (duck-typed by attribute)."""
    sel1 = np.asarray(select_all.grid_mapping, dtype=np.int64)          # raw column of each mid channel
    sel2 = np.asarray(select_sub.speech_grid_mapping, dtype=np.int64)   # mid channel of each output channel
    grid_of_mid = np.full(len(sel1), -1, dtype=np.int64)
    comp_lists = []
    for g, (used, applied) in enumerate(zip(car.selection_masks_computation, car.selection_masks_application)):
        grid_of_mid[np.nonzero(applied)[0]] = g
        comp_lists.append(sel1[np.nonzero(used)[0]])                    # ascending mid index = numpy's summation order
    return sel1[sel2], grid_of_mid[sel2], comp_lists
