"""Pin the CPU oracle's HGA half against the golden vectors produced by the REFERENCE's own Cython
module + scipy (oracle/make_golden.py).  Bit-exact (float64 array_equal), no tolerance."""
import hashlib

import numpy as np

from dss_amd.synthetic import synthetic_ecog


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def test_sosfilt_matches_scipy_bitwise(oracle, golden):
    from scipy.signal import sosfilt
    f = golden("hga_filters.npz")
    x = synthetic_ecog(11, 333, 7)
    zi = np.repeat(f["zi_hg"][:, :, None], 7, axis=2)
    want, zf = sosfilt(f["sos_hg"], x, axis=0, zi=zi)
    got, zg = oracle.sosfilt(f["sos_hg"], x, zi)
    assert np.array_equal(got, want) and np.array_equal(zg, zf)
    # chunked == whole (state carry)
    a, z1 = oracle.sosfilt(f["sos_hg"], x[:100], zi)
    b, z2 = oracle.sosfilt(f["sos_hg"], x[100:], z1)
    assert np.array_equal(np.concatenate([a, b]), want) and np.array_equal(z2, zf)


def test_small_case_stored_input(oracle, golden):
    g, f = golden("hga_frames.npz"), golden("hga_filters.npz")
    got = oracle.extractor(f, 8).extract(g["small_in"])
    assert got.shape == g["small_out"].shape == (16, 8)
    assert np.array_equal(got, g["small_out"])


def test_offline_trials_100_frames(oracle, golden):
    g, f = golden("hga_frames.npz"), golden("hga_filters.npz")
    for b in range(4):
        x = synthetic_ecog(1000 + b, 1040, 64)
        assert np.array_equal(_sha(x), g[f"offline{b}_in_sha"]), "synthetic input drifted"
        got = oracle.extractor(f, 64).extract(x)
        assert got.shape == (100, 64)
        assert np.array_equal(got, g[f"offline{b}_out"])


def test_online_40_sample_packets(oracle, golden):
    g, f = golden("hga_frames.npz"), golden("hga_filters.npz")
    x = synthetic_ecog(2000, 1040, 64)
    ex = oracle.extractor(f, 64)
    frames = [ex.extract(x[i:i + 40]) for i in range(0, 1040, 40)]
    assert [len(fr) for fr in frames] == g["online_counts"].tolist() == [1] + [4] * 25
    assert np.array_equal(np.concatenate(frames), g["online_out"])


def test_ragged_packets_case1_first(oracle, golden):
    g, f = golden("hga_frames.npz"), golden("hga_filters.npz")
    sizes = g["ragged_sizes"].tolist()
    x = synthetic_ecog(2001, sum(sizes), 5)
    ex = oracle.extractor(f, 5)
    frames, pos = [], 0
    for s in sizes:
        frames.append(ex.extract(x[pos:pos + s]))
        pos += s
    assert [len(fr) for fr in frames] == g["ragged_counts"].tolist()
    assert np.array_equal(np.concatenate(frames), g["ragged_out"])


def test_framebuffer_and_log_power_alone(oracle, golden):
    g = golden("hga_frames.npz")
    x = g["rawfb_in"]
    fb = oracle.framebuffer(0.05, 0.01, 1000, 3)
    parts = [oracle.log_power(fb.insert(x[a:b])) for a, b in ((0, 30), (30, 100), (100, 300))]
    assert np.array_equal(np.concatenate(parts), g["rawfb_out"])
    # reset() returns to warm-start behaviour
    fb.reset()
    assert np.array_equal(oracle.log_power(fb.insert(x[0:30])), g["rawfb_out"][:1])


def test_log_is_log_of_mean_power(oracle):
    x = synthetic_ecog(5, 120, 4)
    assert np.array_equal(np.log(oracle.log_power(x, mean_only=True)), oracle.log_power(x))
