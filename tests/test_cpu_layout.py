"""CPU tests (no GPU): the host-side layout of the CU-resident sample kernel -- which lane runs which unit, register
slots, h-gate image, tail records and column tables -- walked exactly as the kernel indexes it and compared, row by
row and block by block, with the model's sparse lists (dss_selftest_fast_layout), next to an independent numpy count of
what the layout should contain."""
import ctypes

import numpy as np
import pytest

from dss_amd import _lib
from dss_amd.lpcnet_weights import make_synthetic_weights, synthetic_blob


def _layout(blob):
    L = _lib.load()
    info = (ctypes.c_int * 8)()
    rc = L.dss_selftest_fast_layout(blob, len(blob), info)
    assert rc == 0, L.dss_last_error().decode()
    keys = ("fast_path", "zr_max", "h_max", "lds_bytes", "zr_cap", "tail_blocks", "mismatches", "oob")
    return dict(zip(keys, list(info)))


def _counts(seed, skew):
    idx = make_synthetic_weights(seed, skew=skew)["gru_a_idx"]
    pos, cnt = 0, []
    for _ in range(144):
        cnt.append(int(idx[pos]))
        pos += 1 + cnt[-1]
    return np.array(cnt).reshape(3, 48)


@pytest.mark.parametrize("seed,skew,fast", [(0, 0.0, 1), (1, 0.0, 1), (2, 0.0, 1), (0, 0.02, 2), (7, 0.05, 2), (0, 0.1, 2),
                                             (3, 0.1, None), (7, 0.3, 0)])
def test_layout_reproduces_every_row(seed, skew, fast):
    info = _layout(synthetic_blob(seed, skew=skew))
    assert info["mismatches"] == 0 and info["oob"] == 0, info
    cnt = _counts(seed, skew)
    zr = np.maximum(cnt[0], cnt[1])
    assert info["zr_max"] == zr.max() and info["h_max"] == cnt[2].max()
    if fast is not None:
        assert info["fast_path"] == fast, info
    if info["fast_path"]:
        assert info["lds_bytes"] <= 151552                 # DSS_HBLK_BYTES (csrc/dss_common.h)
        # what must sit in LDS at least: every h block, and every z/r block beyond the register slots
        order = np.argsort(-zr, kind="stable")
        caps = np.empty(48, int)
        caps[order[:16]] = info["zr_cap"]
        caps[order[16:]] = 8
        tails = int(np.maximum(cnt[0] - caps, 0).sum() + np.maximum(cnt[1] - caps, 0).sum())
        assert info["tail_blocks"] == tails, (info, tails)
        assert info["lds_bytes"] >= 128 * (cnt[2].sum() + tails)
        assert (info["fast_path"] == 2) == bool(tails or cnt[2].max() > 28)
        assert info["zr_cap"] == (12 if (info["fast_path"] == 1 and zr.max() > 10) else 10)


def test_seeded_model_image_is_tight():
    """Round 2: the h-gate image holds each row group's own list (no padding to the wave's longest): 922 blocks of the
    seeded model + alignment spares in 121 KB, where the padded form needed 135 KB."""
    info = _layout(synthetic_blob(0))
    assert info["fast_path"] == 1 and info["tail_blocks"] == 0
    assert 922 * 128 <= info["lds_bytes"] <= 922 * 128 + 48 * 128


def test_layout_rejects_garbage():
    L = _lib.load()
    info = (ctypes.c_int * 8)()
    assert L.dss_selftest_fast_layout(b"\0" * 16, 16, info) != 0
    blob = bytearray(synthetic_blob(0))
    blob[0:4] = b"XXXX"
    assert L.dss_selftest_fast_layout(bytes(blob), len(blob), info) != 0


def test_converter_reports_the_kernel_fit():
    from dss_amd.nnet_data import kernel_fit
    assert kernel_fit(synthetic_blob(0))["fast_path"] == 1
    fit = kernel_fit(synthetic_blob(7, skew=0.05))
    assert fit["fast_path"] == 2 and "tail" in fit["kernel"] and fit["mismatches"] == 0
    assert kernel_fit(synthetic_blob(7, skew=0.3))["fast_path"] == 0
