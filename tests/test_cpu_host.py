"""CPU tests of the host-side code around the hot path: the PyTorch model classes against golden vectors produced by
the reference's own local/models.py, the wire/disk formats, the electrode tables and the fused
front-end description against the CPU chain in oracle/ecog_chain_oracle.py, the shipped filter tables, and the runner
that swaps the GPU units into a user's unchanged reference script."""
import hashlib
import os
import struct
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def test_models_match_reference_golden(golden):
    from dss_amd.models import BidirectionalSpeechSynthesisModel, UnidirectionalVoiceActivityDetector
    g = golden("models.npz")
    torch.manual_seed(0)
    m = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64).eval()
    assert sum(p.numel() for p in m.parameters()) == int(g["bilstm_params"][0]) == 378420
    sd = m.state_dict()
    assert set(sd) >= {"lstm.weight_ih_l0", "lstm.weight_hh_l1_reverse", "regressor.weight", "regressor.bias"}
    flat = np.concatenate([v.numpy().ravel() for _, v in sorted(sd.items())])
    assert np.array_equal(_sha(flat), g["bilstm_sd_sha"])            # same init stream => same checkpoint layout
    with torch.no_grad():
        y, state = m(torch.from_numpy(g["bilstm_in"]), m.create_new_initial_state(batch_size=1))
    assert y.shape == (1, 100, 20) and state[0].shape == (4, 1, 100)
    np.testing.assert_allclose(y.numpy(), g["bilstm_out"], rtol=0, atol=1e-6)
    torch.manual_seed(1)
    v = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=64).eval()
    assert sum(p.numel() for p in v.parameters()) == int(g["vad_params"][0]) == 311102
    x = torch.from_numpy(g["bilstm_in"])
    with torch.no_grad():
        s = v.create_new_initial_state(batch_size=1)
        y1, s = v(x[:, :4], s)
        y2, s = v(x[:, 4:8], s)
        y_none, _ = v(x[:, :4])                                        # state=None -> zero state
    np.testing.assert_allclose(np.concatenate([y1.numpy(), y2.numpy()], axis=1), g["vad_out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(y_none.numpy(), y1.numpy(), rtol=0, atol=0)


def test_wire_and_disk_formats(tmp_path):
    from dss_amd import formats as F
    assert F.PACKET_TOPIC == bytes([4, 1, 2]) and F.PACKET_HEADER.size == 7
    samples = np.arange(129 * 40, dtype=np.float32).reshape(129, 40)               # channel-major on the wire
    pkt = struct.pack("=BBB HH", 4, 1, 2, 129, 40) + samples.tobytes()             # development_amplifier.py:14-25
    arr = F.parse_packet(pkt)
    assert arr.shape == (40, 129) and arr.dtype == np.float64 and arr.flags["C_CONTIGUOUS"]
    assert np.array_equal(arr, samples.T.astype(np.float64))
    assert F.build_packet(arr) == pkt and pkt[:3] == F.PACKET_TOPIC
    with pytest.raises(ValueError):
        F.parse_packet(b"\x04\x01")
    # stream logs: BinaryLogger appends message.data.tobytes() (units.py:264-270)
    rng = np.random.default_rng(0)
    hga = [rng.standard_normal((4, 64)) for _ in range(5)]
    lpc = [rng.standard_normal((n, 20)).astype(np.float32) for n in (100, 37)]
    with open(tmp_path / "log.hga.f64", "wb") as fh:
        for a in hga:
            F.append_stream_log(fh, a)
    with open(tmp_path / "log.lpc.f32", "wb") as fh:
        for a in lpc:
            F.append_stream_log(fh, a)
    assert np.array_equal(F.read_stream_log(tmp_path / "log.hga.f64", 64, np.float64), np.concatenate(hga))
    assert np.array_equal(F.read_stream_log(tmp_path / "log.lpc.f32", 20, np.float32), np.concatenate(lpc))
    with pytest.raises(ValueError):
        F.read_stream_log(tmp_path / "log.lpc.f32", 64, np.float32)
    assert sum(1 for _ in F.iter_packets(np.zeros((130, 129)), 40)) == 3
    # VAD label file (units.py:311-319) and feature files (LPCNet.pyx:90-115)
    (tmp_path / "log.vad.lab").write_text(F.format_vad_label(250, 100) + F.format_vad_label(1000, 7))
    assert (tmp_path / "log.vad.lab").read_text().splitlines()[0] == "2.50\t3.50\t100 frames"
    assert F.read_vad_labels(tmp_path / "log.vad.lab") == [(2.5, 3.5, 100), (10.0, 10.07, 7)]
    feats36 = rng.standard_normal((9, 36)).astype(np.float32)
    feats36.tofile(tmp_path / "utt.f32")
    assert np.array_equal(F.read_feature_file(tmp_path / "utt.f32"), feats36[:, :20])
    import LPCNet
    assert np.array_equal(np.stack(list(LPCNet.LPCFeatureFile(str(tmp_path / "utt.f32")))), feats36[:, :20])
    assert F.pcm_to_s16le(np.array([1, -2, 32767], dtype=np.int16)) == struct.pack("<3h", 1, -2, 32767)
    # the packet body as it is (what the device-side ingest takes): a float32 view, channel-major, no copy
    pk = F.build_packet(np.arange(12, dtype=np.float64).reshape(4, 3))
    body = F.packet_payload(pk)
    assert body.dtype == np.float32 and body.shape == (3, 4) and not body.flags.owndata
    assert np.array_equal(body.T.astype(np.float64), F.parse_packet(pk))


def test_electrode_tables_and_frontend_description():
    from dss_amd import electrodes as E
    from ecog_chain_oracle import ZScore, reference_chain
    assert len(E.GRID_COLUMNS) == 128 and sorted(E.GRID_COLUMNS) == list(range(128))
    keep = E.speech_channels_zero_based()
    assert len(keep) == 64 and not set(keep + 1) & set(E.BAD_CHANNELS)
    both, car, speech = reference_chain()
    # the oracle chain against closed forms (its only pins: the reference has no test for these classes)
    x = np.random.default_rng(0).standard_normal((7, 129))
    y = both(x)
    assert y.shape == (7, 128) and np.array_equal(y[:, 0], x[:, 125])
    out = car(y)
    ok = np.ones(64, dtype=bool)
    ok[[18, 37, 47, 51]] = False                                   # channels 19, 38, 48, 52 stay out of the mean
    np.testing.assert_allclose(out[:, :64], y[:, :64] - y[:, :64][:, ok].mean(axis=1, keepdims=True), atol=1e-12)
    np.testing.assert_allclose(out[:, 64:], y[:, 64:] - y[:, 64:].mean(axis=1, keepdims=True), atol=1e-12)
    z = ZScore(np.full((1, 64), 2.0), np.full((1, 64), 4.0))(speech(out))
    assert z.shape == (7, 64) and np.allclose(z, (speech(out) - 2.0) / 4.0)
    # the front-end description built from the tables alone equals the one derived from the chain objects ...
    a = E.reference_frontend()
    b = E.frontend_from_transforms(both, car, speech)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert all(np.array_equal(p, q) for p, q in zip(a[2], b[2])) and [len(c) for c in a[2]] == [60, 64]
    # ... and describes the chain: out[:, c] = raw[:, src_col[c]] - sequential mean over comp_lists[grid_of[c]]
    src, gof, comp = a
    want = speech(car(both(x)))
    got = np.empty_like(want)
    for c in range(64):
        acc = np.zeros(7)
        for col in comp[gof[c]]:
            acc = acc + x[:, col]
        got[:, c] = x[:, src[c]] - acc / len(comp[gof[c]])
    assert np.array_equal(got, want)                                # bit-identical, including the summation order


def test_transform_chain_against_the_reference_classes(golden):
    """tests/golden/ecog_chain.npz = every stage of decode_online.py:65-97's chain as the reference's OWN classes computed it
    (SelectElectrodesFromBothGrids, CommonAverageReferencing, SelectElectrodesOverSpeechAreas, ZScoreNormalization of
    /root/reference/local/common.py, objects built as configure_feature_transforms builds them; oracle/make_golden.py).  The
    CPU restatement, the shipped channel tables and the GPU front end's description reproduce it bit for bit."""
    from dss_amd import electrodes as E
    from ecog_chain_oracle import ZScore, reference_chain
    g = golden("ecog_chain.npz")
    both, car, speech = reference_chain()
    raw = g["raw"]
    assert raw.shape == (80, 129)
    assert np.array_equal(np.asarray(both.grid_mapping), g["grid_mapping"]) and tuple(g["grid_mapping"]) == E.GRID_COLUMNS
    assert np.array_equal(speech.speech_grid_mapping, g["speech_grid_mapping"])
    assert np.array_equal(E.speech_channels_zero_based(), g["speech_grid_mapping"]) and tuple(g["exclude_channels"]) == E.BAD_CHANNELS
    assert np.array_equal(np.stack(car.selection_masks_application), g["masks_application"])
    assert np.array_equal(np.stack(car.selection_masks_computation), g["masks_computation"])
    a = both(raw)
    b = car(a)
    c = speech(b)
    assert np.array_equal(a, g["after_select_both"]) and np.array_equal(b, g["after_car"]) and np.array_equal(c, g["after_select_speech"])
    # the front end's description (what csrc/hga_kernels.hip hga_frontend_kernel executes): a sequential sum per grid
    src, gof, comp = E.reference_frontend()
    got = np.empty_like(c)
    for ch in range(64):
        acc = np.zeros(len(raw))
        for col in comp[gof[ch]]:
            acc = acc + raw[:, col]
        got[:, ch] = raw[:, src[ch]] - acc / len(comp[gof[ch]])
    assert np.array_equal(got, g["after_select_speech"])
    # ZScoreNormalization, statistics selected as decode_online.py:94-95 selects them
    zs = ZScore(speech(g["zs_means_128"].reshape((1, -1))), speech(g["zs_stds_128"].reshape((1, -1))))
    assert np.array_equal(zs.channel_means, g["zs_means"]) and np.array_equal(zs.channel_stds, g["zs_stds"])
    assert np.array_equal(zs(g["zs_in"]), g["zs_out"])
    assert np.array_equal((g["zs_in"] - g["zs_means"]) / g["zs_stds"], g["zs_out"])      # the two IEEE operations dss_hga_set_zscore applies


def test_shipped_filter_tables_are_what_scipy_designs_here(golden):
    from dss_amd import hga
    g = golden("hga_filters.npz")
    tab = hga.reference_filters(1000, 70, 170)
    for a, k in zip(tab, ("sos_hg", "sos_fh", "zi_hg", "zi_fh")):
        assert np.array_equal(a, g[k])                              # the tables the golden HGA frames were made with
    for a, b in zip(hga.design_filters(1000, 70, 170), tab):        # scipy in this image still reproduces them
        assert np.array_equal(a, b)
    other = hga.reference_filters(2000, 70, 170)                    # any other configuration is designed on the spot
    assert other[0].shape == (8, 6) and not np.array_equal(other[0], tab[0])


def test_runner_swaps_the_gpu_units_into_an_unchanged_user_script(tmp_path):
    """python -m dss_amd.run <script>: the user's own local.units keeps every class except the accelerated ones."""
    (tmp_path / "local").mkdir()
    (tmp_path / "local" / "__init__.py").write_text("")
    (tmp_path / "local" / "units.py").write_text(textwrap.dedent("""
        import LPCNet                       # resolves to the drop-in module
        from hga_optimized import WarmStartFrameBuffer
        class HighGammaExtractor: origin = 'user'
        class DelayedLPCNetVocoder: origin = 'user'
        class RecurrentNeuralDecodingModel: origin = 'user'
        class BinaryLogger: origin = 'user'
        class HighGammaActivity:
            def make(self):
                return HighGammaExtractor   # looked up at call time, like units.py:199-201
    """))
    (tmp_path / "script.py").write_text(textwrap.dedent("""
        import sys
        from local.units import HighGammaActivity, DelayedLPCNetVocoder, RecurrentNeuralDecodingModel, BinaryLogger
        import LPCNet
        print('ARGS', sys.argv[1:])
        print('EXT', HighGammaActivity().make().__module__)
        print('VOC', DelayedLPCNetVocoder.__module__)
        print('DEC', RecurrentNeuralDecodingModel.__module__)
        print('LOG', BinaryLogger.origin)
        print('LPCNET', LPCNet.__file__)
    """))
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "delayed-speech-synthesis_amd"))
    out = subprocess.run([sys.executable, "-m", "dss_amd.run", str(tmp_path / "script.py"), "cfg.ini", "--run"],
                         env=env, text=True, capture_output=True, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    assert "ARGS ['cfg.ini', '--run']" in out.stdout
    assert "EXT dss_amd.units" in out.stdout and "VOC dss_amd.units" in out.stdout and "LOG user" in out.stdout
    assert "DEC dss_amd.units" in out.stdout and "local.units.RecurrentNeuralDecodingModel: replaced" in out.stderr
    assert os.path.join("delayed-speech-synthesis_amd", "LPCNet.py") in out.stdout
    assert "local.units.HighGammaExtractor: replaced" in out.stderr
    assert "local.training.AsynchronousSynthesisQueue: left alone" in out.stderr      # this user tree has no training.py


def test_kernel_fit_checks_of_the_recurrent_models():
    """Which modules the library's own LSTM kernels take over (host-side check, no GPU): exactly the reference's two
    architectures within the kernels' sizes; anything else stays a PyTorch-ROCm module."""
    from dss_amd import decoder, vad
    from dss_amd.models import BidirectionalSpeechSynthesisModel, UnidirectionalVoiceActivityDetector
    dec = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64)
    det = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=64)
    assert decoder.fits(dec) and vad.fits(det)
    assert not decoder.fits(det) and not vad.fits(dec)                      # the other architecture's state_dict
    assert not decoder.fits(BidirectionalSpeechSynthesisModel(nb_layer=1, nb_hidden_units=100, nb_electrodes=64))
    assert not decoder.fits(BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=129, nb_electrodes=64))
    assert not decoder.fits(BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=257))
    assert decoder.fits(BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=128, nb_electrodes=256))
    assert not decoder.fits(object())
    wide = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64)
    wide.regressor = torch.nn.Linear(200, 40)                               # more outputs than the regressor kernel stages in LDS
    assert not decoder.fits(wide)
    assert len(decoder._KEYS) == 18 and decoder._KEYS[0] == "lstm.weight_ih_l0" and decoder._KEYS[4] == "lstm.weight_ih_l0_reverse"
    assert list(dec.state_dict().keys()) == list(decoder._KEYS)             # the C ABI takes the arrays in state_dict order


def test_decoder_unit_on_the_host_path(golden, tmp_path):
    """RecurrentNeuralDecodingModel without a GPU: built from settings and a state_dict file like the reference's unit
    (units.py:481-497), one segment in, (L, 20) features at 100 Hz out, equal to the reference-built golden vector."""
    import asyncio
    import dss_amd.units as U
    from dss_amd.models import BidirectionalSpeechSynthesisModel
    g = golden("models.npz")
    torch.manual_seed(0)
    ref = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64)
    path = tmp_path / "decoder.pth"
    torch.save(ref.state_dict(), path)
    unit = U.RecurrentNeuralDecodingModel(U.RecurrentNeuralDecodingModelSettings(
        path_to_model_weights=str(path), model=BidirectionalSpeechSynthesisModel,
        params=dict(nb_layer=2, nb_hidden_units=100, nb_electrodes=64)))
    unit.initialize()

    async def drive(gen):
        return [item async for item in gen]
    seg = g["bilstm_in"][0].astype(np.float64)             # the unit casts to float32 itself (units.py:503)
    for _ in range(2):                                     # a fresh state per segment: the same answer twice
        (stream, msg), = asyncio.run(drive(unit.decode(U.ClosedLoopMessage(data=seg, fs=100, previous_frames=7))))
        assert stream is unit.OUTPUT and msg.fs == 100 and msg.previous_frames == 7
        assert msg.data.shape == (100, 20) and msg.data.dtype == np.float32
        np.testing.assert_allclose(msg.data, g["bilstm_out"][0], rtol=0, atol=2e-5)
    if not torch.cuda.is_available():
        assert unit.STATE.kernel is None and unit.STATE.device == "cpu"
