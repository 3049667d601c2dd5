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
    # every xiph symbol the reference's cLPCNet.pxd:10-19 binds: decoder (implemented) and encoder (failing stubs)
    for n in ("lpcnet_create", "lpcnet_init", "lpcnet_destroy", "lpcnet_synthesize", "lpcnet_encoder_create",
              "lpcnet_encoder_init", "lpcnet_encoder_destroy", "lpcnet_compute_features",
              "lpcnet_compute_single_frame_features"):
        assert n in names
    assert set(_lib.EXPORTED_SYMBOLS) == set(names)


REF_PYX = "/root/reference/extensions/lpcnet/LPCNet.pyx"


@pytest.mark.skipif(not os.path.exists(REF_PYX), reason="reference tree not present (GPU box)")
def test_reference_pyx_links_against_the_library(tmp_path):
    """The reference's OWN Cython wrapper (extensions/lpcnet/LPCNet.pyx + cLPCNet.pxd, read where they lie) is
    cythonized, compiled against include/compat and linked to libdss_hip.so, unchanged; then imported.  Compile + link
    + import only: no compute call (no GPU here)."""
    import subprocess, sys, sysconfig
    from dss_amd import _lib
    _lib.load()
    pkg = os.path.dirname(_lib.lib_path())
    c_file = tmp_path / "LPCNet.c"
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "-I", os.path.dirname(REF_PYX), "-o", str(c_file), REF_PYX])
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so = tmp_path / ("LPCNet" + ext)
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                           "-I", os.path.join(ROOT, "include", "compat"), "-I", sysconfig.get_paths()["include"],
                           "-I", np.get_include(), "-o", str(so), str(c_file), "-L", pkg, "-ldss_hip",
                           "-Wl,-rpath," + pkg])
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import LPCNet as M\n"
        "assert M.__file__.endswith(%r), M.__file__\n"
        "assert M.LPCNet.LPCNET_FRAME_SIZE == 160 and M.LPCFeatureEncoder.NB_TOTAL_FEATURES == 36\n"
        "for cls in (M.LPCFeatureEncoder, M.LPCNet):\n"          # encoder: stub -> NULL; decoder: no GPU / no weights -> NULL
        "    try:\n"
        "        cls()\n"
        "    except MemoryError:\n"
        "        print('MemoryError', cls.__name__)\n"
        "print('imported')\n" % (str(tmp_path), ext))
    env = {k: v for k, v in os.environ.items() if k not in ("DSS_LPCNET_WEIGHTS", "PYTHONPATH")}
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True)
    assert "imported" in out and "MemoryError LPCFeatureEncoder" in out and "MemoryError LPCNet" in out
    # every xiph symbol the generated C references resolves inside libdss_hip.so
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", str(so)], text=True)
    wanted = sorted(set(re.findall(r"\bU (lpcnet_\w+)", undefined)))
    assert wanted == ["lpcnet_compute_single_frame_features", "lpcnet_create", "lpcnet_destroy", "lpcnet_encoder_create",
                      "lpcnet_encoder_destroy", "lpcnet_encoder_init", "lpcnet_init", "lpcnet_synthesize"], wanted
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.lib_path()], text=True)
    for sym in wanted:
        assert re.search(r"\bT %s\b" % sym, exported), sym


def test_void_synthesize_fails_soft_and_counts():
    """lpcnet_synthesize has no error channel (cLPCNet.pxd:13) and runs inside the live decode loop: a failure must
    neither abort the process nor leave the caller's np.ones(160) buffer (LPCNet.pyx:37) as audio."""
    import ctypes
    from dss_amd import _lib
    L = _lib.load()
    out = np.ones(160, dtype=np.int16)
    feats = np.zeros(20, dtype=np.float32)
    before = L.dss_error_count()
    L.lpcnet_synthesize(None, feats.ctypes.data, out.ctypes.data, 160)          # NULL state
    assert not out.any() and L.dss_error_count() == before + 1
    assert b"null" in L.dss_last_error().lower()
    assert L.lpcnet_encoder_create() is None and b"encoder" in L.dss_last_error().lower()
    f36 = np.ones(36, np.float32)
    assert L.lpcnet_compute_single_frame_features(None, out.ctypes.data, f36.ctypes.data) == -1 and not f36.any()


def test_no_weights_is_an_error_not_random_noise():
    """LPCNet() must not fall back to the seeded synthetic model silently."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import LPCNet\n"
            "try:\n"
            "    LPCNet.LPCNet()\n"
            "except MemoryError as e:\n"
            "    print('MemoryError:', e)\n" % os.path.join(ROOT, "delayed-speech-synthesis_amd"))
    env = {k: v for k, v in os.environ.items() if k not in ("DSS_LPCNET_WEIGHTS", "DSS_LPCNET_SYNTHETIC")}
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True)
    assert "MemoryError: no LPCNet weights" in out and "DSS_LPCNET_WEIGHTS" in out


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
