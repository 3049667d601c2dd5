"""GPU parity of the HGA path, through the C ABI, against (a) the golden vectors produced by the reference's
own Cython module + scipy and (b) the CPU oracle.  float64, bit-exact (array_equal) on the host-buffer path;
the device-resident log variant is held to <= 1 ulp (stated in DESIGN.md)."""
import numpy as np
import pytest

from dss_amd.synthetic import synthetic_ecog

pytestmark = pytest.mark.gpu


def _filters(golden):
    f = golden("hga_filters.npz")
    return f["sos_hg"], f["sos_fh"], f["zi_hg"], f["zi_fh"]


def test_small_case_stored_input(golden):
    from dss_amd.hga import HgaExtractorGPU
    g = golden("hga_frames.npz")
    ex = HgaExtractorGPU(1, 8, filters=_filters(golden))
    got = ex.extract(g["small_in"])[0]
    assert got.shape == (16, 8)
    assert np.array_equal(got, g["small_out"])


def test_offline_trials_single_and_multi_stream(golden):
    from dss_amd.hga import HgaExtractorGPU
    g = golden("hga_frames.npz")
    xs = np.stack([synthetic_ecog(1000 + b, 1040, 64) for b in range(4)])
    ex = HgaExtractorGPU(4, 64, filters=_filters(golden))
    got = ex.extract(xs)
    assert got.shape == (4, 100, 64)
    for b in range(4):
        assert np.array_equal(got[b], g[f"offline{b}_out"])
    # a fresh single-stream extractor gives the same frames (streams are independent)
    one = HgaExtractorGPU(1, 64, filters=_filters(golden)).extract(xs[2])[0]
    assert np.array_equal(one, g["offline2_out"])


def test_online_packets_state_carry_and_reset(golden):
    from dss_amd.hga import HgaExtractorGPU
    g = golden("hga_frames.npz")
    x = synthetic_ecog(2000, 1040, 64)
    ex = HgaExtractorGPU(1, 64, filters=_filters(golden))
    frames = [ex.extract(x[i:i + 40])[0] for i in range(0, 1040, 40)]
    assert [len(f) for f in frames] == g["online_counts"].tolist()
    assert np.array_equal(np.concatenate(frames), g["online_out"])
    ex.reset()
    again = [ex.extract(x[i:i + 40])[0] for i in range(0, 200, 40)]
    assert np.array_equal(np.concatenate(again), g["online_out"][:17])


def test_ragged_packets_odd_channels(golden):
    from dss_amd.hga import HgaExtractorGPU
    g = golden("hga_frames.npz")
    sizes = g["ragged_sizes"].tolist()
    x = synthetic_ecog(2001, sum(sizes), 5)
    ex = HgaExtractorGPU(1, 5, filters=_filters(golden))
    frames, pos = [], 0
    for s in sizes:
        assert ex.frames_for(s) >= 0
        frames.append(ex.extract(x[pos:pos + s])[0])
        pos += s
    assert [len(f) for f in frames] == g["ragged_counts"].tolist()
    assert np.array_equal(np.concatenate(frames), g["ragged_out"])


def test_dropin_module_compute_log_power_features(golden):
    import hga_optimized as dropin
    g = golden("hga_frames.npz")
    x = g["rawfb_in"]
    fb = dropin.WarmStartFrameBuffer(frame_length=0.05, frame_shift=0.01, fs=1000, nb_channels=3)
    parts = [dropin.compute_log_power_features(fb.insert(x[a:b].copy()), 1000, 0.05, 0.01)
             for a, b in ((0, 30), (30, 100), (100, 300))]
    assert np.array_equal(np.concatenate(parts), g["rawfb_out"])
    with pytest.raises(ValueError):
        dropin.compute_log_power_features(x.astype(np.float32), 1000, 0.05, 0.01)
    # too short for a single window -> empty result, like the reference (num_windows <= 0)
    assert dropin.compute_log_power_features(x[:20].copy(), 1000, 0.05, 0.01).shape[0] == 0


def test_device_resident_variant_and_mean_power_bit_exact(oracle, golden):
    import torch
    from dss_amd.hga import HgaExtractorGPU
    x = synthetic_ecog(1001, 1040, 64)
    want_log = golden("hga_frames.npz")["offline1_out"]
    ex = HgaExtractorGPU(1, 64, filters=_filters(golden))
    d = torch.from_numpy(x[None]).cuda()
    p = ex.extract_torch(d, apply_log=False).cpu().numpy()[0]
    assert np.array_equal(np.log(p), want_log)                  # mean power + 0.01 is bit-exact; log by host libm
    ex.reset()
    q = ex.extract_torch(d, apply_log=True).cpu().numpy()[0]
    ulp = np.abs(q - want_log) / np.spacing(np.abs(want_log))
    assert ulp.max() <= 1.0                                     # OCML log vs glibc log: tolerance 1 ulp


def test_full_size_properties_128_streams():
    """Config-5 size (128 streams x 64 ch): stream independence and chunking invariance."""
    from dss_amd.hga import HgaExtractorGPU
    S = 128
    xs = np.stack([synthetic_ecog(5000 + s, 400, 64) for s in range(S)])
    whole = HgaExtractorGPU(S, 64).extract(xs)                  # one CASE-1 chunk
    ex = HgaExtractorGPU(S, 64)
    parts = [ex.extract(xs[:, a:b]) for a, b in ((0, 80), (80, 120), (120, 400))]
    assert np.array_equal(np.concatenate(parts, axis=1), whole)
    perm = np.random.default_rng(0).permutation(S)
    shuffled = HgaExtractorGPU(S, 64).extract(xs[perm])
    assert np.array_equal(shuffled, whole[perm])
    assert np.isfinite(whole).all() and whole.shape == (S, 36, 64)


@pytest.mark.parametrize("fs,wl,ws,packets", [(8000, 0.05, 0.01, (900, 333, 1200)),      # ring would need 1024 rows: three-launch form
                                              (2000, 0.05, 0.01, (64, 100, 40, 256, 31)),   # 100-sample windows, ring of 256 rows: fused
                                              (1000, 0.025, 0.005, (12, 40, 7, 90))])       # short windows, first packet < one frame
def test_other_window_shapes_against_the_oracle(oracle, fs, wl, ws, packets):
    """Window shapes other than the reference's 50 ms / 10 ms at 1 kHz, bit-exact against the CPU oracle: they exercise the
    ring sizing of the fused kernel and the fallback to the three-launch form (filter, window, overlap kernels)."""
    from dss_amd.hga import HgaExtractorGPU, design_filters
    hg, fh, zi_hg, zi_fh = design_filters(fs)
    C = 6
    x = synthetic_ecog(3000 + fs, sum(packets), C, fs=fs)
    gpu = HgaExtractorGPU(1, C, fs=fs, window_length=wl, window_shift=ws, filters=(hg, fh, zi_hg, zi_fh))
    cpu = oracle.extractor({"sos_hg": hg, "sos_fh": fh, "zi_hg": zi_hg, "zi_fh": zi_fh}, C, fs=fs, wl=wl, ws=ws)
    pos, total = 0, 0
    for n in packets:
        got = gpu.extract(x[pos:pos + n])[0]
        want = cpu.extract(x[pos:pos + n])
        assert got.shape == want.shape and np.array_equal(got, want), (fs, n, got.shape, want.shape)
        pos += n
        total += len(want)
    assert total > 0


def test_streamed_kernel_equals_the_other_forms(golden):
    """The two forms of the extractor -- hga_fused_kernel (default) and the three launches (the fallback for window shapes
    whose ring does not fit LDS) -- give the same bits: offline trials, 40-sample packets, ragged packets (tiles that end
    inside a packet, packets shorter than a tile, a first packet shorter than a frame)."""
    import torch
    from dss_amd.hga import HgaExtractorGPU
    S = 11                                                     # not a multiple of 8: exercises the block -> stream mapping's tail
    xs = np.stack([synthetic_ecog(6000 + s, 1040, 64) for s in range(S)])
    d = torch.from_numpy(xs).cuda()
    res = {}
    for path in (0, 2):
        ex = HgaExtractorGPU(S, 64, filters=_filters(golden))
        ex._force_path(path)
        whole = ex.extract_torch(d, apply_log=False).cpu().numpy()
        ex.reset()
        parts, pos = [], 0
        for n in (13, 40, 40, 17, 100, 64, 23, 200, 543):
            parts.append(ex.extract_torch(d[:, pos:pos + n].contiguous(), apply_log=False).cpu().numpy())
            pos += n
        res[path] = (whole, np.concatenate(parts, axis=1))
    assert res[0][0].shape == (S, 100, 64)
    assert np.array_equal(res[0][0], res[2][0])
    assert np.array_equal(res[0][1], res[2][1])
    g = golden("hga_frames.npz")                               # and the reference's own frames, through the three launches
    ex = HgaExtractorGPU(4, 64, filters=_filters(golden))
    ex._force_path(2)
    got = ex.extract(np.stack([synthetic_ecog(1000 + b, 1040, 64) for b in range(4)]))
    for b in range(4):
        assert np.array_equal(got[b], g[f"offline{b}_out"])


def test_raw_packets_and_zscore_in_one_launch(golden):
    """raw 129-column packets -> reorder + per-grid CAR + select -> filters -> log power -> z-score: the
    front end, then the extractor with the z-score as its epilogue -- hga_fused_kernel's (default) or, in the three-launch
    form, hga_window_kernel's.  Both equal, bit for bit, a host / torch z-score of the plain frames; odd packet lengths
    make the raw rows start on 8-byte boundaries."""
    import torch
    from dss_amd.electrodes import reference_frontend
    from dss_amd.hga import HgaExtractorGPU
    S = 9
    rng = np.random.default_rng(3)
    raw = rng.standard_normal((S, 677, 129)) * 40.0
    mean, std = rng.standard_normal(64), rng.uniform(0.5, 2.0, 64)
    d = torch.from_numpy(raw).cuda()
    outs = {}
    for path in (0, 2):
        ex = HgaExtractorGPU(S, 64, filters=_filters(golden))
        ex.set_frontend(129, *reference_frontend())
        ex._force_path(path)
        parts, pos = [], 0
        for n in (77, 40, 41, 319, 200):
            parts.append(ex.extract_raw_torch(d[:, pos:pos + n].contiguous(), apply_log=False).cpu().numpy())
            pos += n
        outs[path] = np.concatenate(parts, axis=1)
    assert outs[0].shape[1] > 50 and np.array_equal(outs[0], outs[2])
    # z-score epilogue: device-resident (log on the device) and host-buffer (glibc log) entry points, both forms
    for path in (0, 2):
        ex = HgaExtractorGPU(S, 64, filters=_filters(golden))
        ex.set_frontend(129, *reference_frontend())
        ex._force_path(path)
        plain = ex.extract_raw_torch(d)                                         # log applied, no z-score
        ex.reset()
        ex.set_zscore(mean, std)
        z = ex.extract_raw_torch(d)
        assert torch.equal(z, (plain - torch.from_numpy(mean).cuda()) / torch.from_numpy(std).cuda()), path
        ex.reset()
        zh = ex.extract_raw(raw)
        ex.reset()
        ex.set_zscore(None)
        ph = ex.extract_raw(raw)
        assert np.array_equal(zh, (ph - mean) / std), path
    # odd channel counts take the z-score too (hga_fused_kernel's epilogue)
    x5 = synthetic_ecog(9, 300, 5)
    e5 = HgaExtractorGPU(1, 5, filters=_filters(golden))
    p5 = e5.extract_torch(torch.from_numpy(x5[None]).cuda())
    e5.reset()
    e5.set_zscore(mean[:5], std[:5])
    z5 = e5.extract_torch(torch.from_numpy(x5[None]).cuda())
    assert torch.equal(z5, (p5 - torch.from_numpy(mean[:5]).cuda()) / torch.from_numpy(std[:5]).cuda())


def test_front_end_and_zscore_against_the_reference_classes(golden):
    """tests/golden/ecog_chain.npz (the reference's own pre/post transform classes, oracle/make_golden.py): raw 129-column
    packets through the GPU front end + extractor give, bit for bit, the frames the extractor gives on the reference's
    `after_select_speech` columns (the plain extractor is pinned by hga_frames.npz), packet by packet with carried state; the
    z-score epilogue with the reference's selected statistics equals ZScoreNormalization's two operations on those frames."""
    from dss_amd.electrodes import reference_frontend
    from dss_amd.hga import HgaExtractorGPU
    g = golden("ecog_chain.npz")
    raw, pre = g["raw"], g["after_select_speech"]
    fe = HgaExtractorGPU(1, 64, filters=_filters(golden))
    fe.set_frontend(129, *reference_frontend())
    plain = HgaExtractorGPU(1, 64, filters=_filters(golden))
    frames = []
    for a in (0, 40):
        got = fe.extract_raw(raw[None, a:a + 40])
        want = plain.extract(pre[None, a:a + 40])
        assert got.shape == want.shape and got.shape[1] == (1 if a == 0 else 4) and np.array_equal(got, want)
        frames.append(want[0])
    frames = np.concatenate(frames)
    z = HgaExtractorGPU(1, 64, filters=_filters(golden))
    z.set_frontend(129, *reference_frontend())
    z.set_zscore(g["zs_means"].reshape(-1), g["zs_stds"].reshape(-1))
    got = np.concatenate([z.extract_raw(raw[None, a:a + 40])[0] for a in (0, 40)])
    assert np.array_equal(got, (frames - g["zs_means"]) / g["zs_stds"])


def test_wire_format_payloads_equal_the_parsed_packets(golden):
    """dss_hga_extract_wire_dev: the bodies of the amplifier's packets as they arrive (float32, channel-major; formats.packet_payload)
    give, bit for bit, the frames of the packets parsed on the host as the reference's ZMQConnector.interpret_bytes parses them
    (formats.parse_packet: reshape, transpose, astype(float64)) -- plain 64-channel packets and raw 129-channel packets through the
    front end, several packets with carried state, an odd packet length."""
    import torch
    from dss_amd import formats as F
    from dss_amd.electrodes import reference_frontend
    from dss_amd.hga import HgaExtractorGPU
    S = 5
    rng = np.random.default_rng(11)
    for c_in, front in ((64, False), (129, True)):
        a = HgaExtractorGPU(S, 64, filters=_filters(golden))
        b = HgaExtractorGPU(S, 64, filters=_filters(golden))
        if front:
            a.set_frontend(129, *reference_frontend())
            b.set_frontend(129, *reference_frontend())
        n_frames = 0
        for n in (40, 40, 37, 40, 120):
            packets = [F.build_packet((rng.standard_normal((n, c_in)) * 40.0)) for _ in range(S)]      # bytes, as on the socket
            parsed = np.stack([F.parse_packet(p) for p in packets])                                    # (S, n, c_in) float64
            payload = np.stack([F.packet_payload(p) for p in packets])                                 # (S, c_in, n) float32
            assert payload.dtype == np.float32 and np.array_equal(payload.transpose(0, 2, 1).astype(np.float64), parsed)
            want = (a.extract_raw_torch if front else a.extract_torch)(torch.from_numpy(parsed).cuda())
            got = b.extract_wire_torch(torch.from_numpy(payload).cuda())
            assert got.shape == want.shape and torch.equal(got, want), (c_in, n)
            n_frames += got.shape[1]
        assert n_frames > 20
    with pytest.raises(ValueError):
        b.extract_wire_torch(torch.zeros((S, 64, 40), dtype=torch.float32, device="cuda"))            # front end set: 129 channels expected
