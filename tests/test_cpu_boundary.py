"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/dss_hip.h declares; host-only logic of the drop-in modules; error behaviour without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dss_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:dss|lpcnet)_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from dss_amd import _lib
    L = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"libdss_hip.so does not export {n}"
    # the four xiph symbols the reference's cLPCNet.pxd:10-13 binds
    for n in ("lpcnet_create", "lpcnet_init", "lpcnet_destroy", "lpcnet_synthesize"):
        assert n in names
    assert set(_lib.EXPORTED_SYMBOLS) == set(names)


def test_num_windows_formula_matches_oracle(oracle):
    from dss_amd import hga
    for T in (10, 49, 50, 51, 59, 60, 80, 140, 1040, 1000):
        assert hga.num_windows(T, 1000, 0.05, 0.01) == oracle.lib.oracle_num_windows(T, 1000, 0.05, 0.01)


def test_blob_roundtrip_and_rejects_garbage():
    from dss_amd import _lib
    from dss_amd.lpcnet_weights import synthetic_blob, unpack_blob, pack_blob, algorithmic_bytes_per_sample
    L = _lib.load()
    blob = synthetic_blob(0)
    dims, w = unpack_blob(blob)
    assert pack_blob(w, dims) == blob
    assert L.dss_lpcnet_load_model(b"garbage" * 20, 140) < 0
    assert b"blob" in L.dss_last_error().lower() or b"dss" in L.dss_last_error().lower()
    assert L.dss_lpcnet_load_model(blob[:-4], len(blob) - 4) < 0          # truncated
    assert L.dss_lpcnet_load_model(blob, len(blob)) == 0
    assert abs(L.dss_lpcnet_bytes_per_sample() - algorithmic_bytes_per_sample(blob)) < 1e-6
    assert 270e3 < L.dss_lpcnet_bytes_per_sample() < 276e3                 # ~273 kB/sample


def test_warm_start_frame_buffer_matches_oracle_and_reference_aliasing(oracle, golden):
    import hga_optimized as dropin
    g = golden("hga_frames.npz")
    x = g["rawfb_in"]
    fb = dropin.WarmStartFrameBuffer(frame_length=0.05, frame_shift=0.01, fs=1000, nb_channels=3)
    ofb = oracle.framebuffer(0.05, 0.01, 1000, 3)
    assert fb.overlap == 40 and fb.frame_length_in_samples == 50
    for a, b in ((0, 30), (30, 100), (100, 300)):
        got = fb.insert(x[a:b].copy())
        assert np.array_equal(got, ofb.insert(x[a:b]))
    # CASE 1 returns the caller's array itself (hga_optimized.pyx:104-107)
    fb.reset()
    chunk = x[:60].copy()
    assert fb.insert(chunk) is chunk
    assert np.shares_memory(fb.remainder_data, chunk)
    with pytest.raises(ValueError):
        fb.insert(x[:20].astype(np.float32))


def test_no_gpu_fails_loudly():
    """There is no CPU fallback: without a device every compute entry point raises."""
    from dss_amd import _lib
    L = _lib.load()
    if L.dss_device_count() > 0:
        pytest.skip("a GPU is present")
    import LPCNet
    from dss_amd.hga import HgaExtractorGPU, log_power
    with pytest.raises(MemoryError):
        LPCNet.LPCNet()
    with pytest.raises(_lib.DssError):
        HgaExtractorGPU(1, 4)
    with pytest.raises(_lib.DssError):
        log_power(np.zeros((60, 2)), 1000, 0.05, 0.01)


def test_product_does_not_import_the_oracle():
    """Nothing under the package may reference oracle/ (the checker)."""
    pkg = os.path.join(ROOT, "delayed-speech-synthesis_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in src and "oracle_" not in src and "oracle/" not in src.replace("oracle/ ", ""), f
