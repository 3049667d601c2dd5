"""GPU tests of the accelerated units (dss_amd.units) and of the device-resident pipelines (configs 3 and 5)."""
import asyncio
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

import numpy as np
import pytest
import torch

from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features
from dss_amd.synthetic import synthetic_ecog

pytestmark = pytest.mark.gpu


async def _drive(gen):
    return [item async for item in gen]


def test_high_gamma_activity_unit_matches_reference_golden(golden):
    import dss_amd.units as U
    g = golden("hga_frames.npz")
    unit = U.HighGammaActivity(U.HighGammaActivitySettings(fs=1000, nb_electrodes=64))
    unit.initialize()
    x = synthetic_ecog(2000, 1040, 64)
    frames = []
    for i in range(0, 1040, 40):
        (stream, msg), = asyncio.run(_drive(unit.process(U.ClosedLoopMessage(data=x[i:i + 40], fs=1000))))
        assert stream is unit.OUTPUT and msg.fs == 100
        frames.append(msg.data)
    assert np.array_equal(np.concatenate(frames), g["online_out"])     # bit-identical to Cython + scipy


def test_extractor_with_pre_and_post_transforms(golden):
    """decode_online.py:65-97 wiring: 128-ch select -> CAR -> 64-ch select, then z-score after the log power."""
    import dss_amd.units as U
    from ecog_chain_oracle import ZScore, reference_chain          # CPU restatement of local/common.py's classes
    both, car, speech = reference_chain()
    post = ZScore(np.full((1, 64), 3.0), np.full((1, 64), 2.0))
    ex = U.HighGammaExtractor(1000, 64, pre_transforms=[both, car, speech], post_transforms=[post])
    raw = synthetic_ecog(77, 200, 129)
    assert ex._fused_pre is not None                       # reorder + CAR + select run in the GPU front-end kernel
    assert ex._fused_post                                  # ... and the z-score inside the library, behind the log
    got = ex.extract_features(raw)
    plain = U.HighGammaExtractor(1000, 64).extract_features(speech(car(both(raw))))       # numpy transforms
    assert got.shape == (16, 64) and np.array_equal(got, (plain - 3.0) / 2.0)            # bit-identical
    # state carries across packets on the fused path too
    more = ex.extract_features(synthetic_ecog(78, 40, 129))
    assert more.shape == (4, 64) and np.isfinite(more).all()
    # a transform chain that is not the reference's falls back to the host transforms
    ex2 = U.HighGammaExtractor(1000, 64, pre_transforms=[both, speech], post_transforms=[post, post])
    assert ex2._fused_pre is None and not ex2._fused_post
    assert np.array_equal(ex2.extract_features(raw), post(post(U.HighGammaExtractor(1000, 64).extract_features(speech(both(raw))))))

    # an object that merely LOOKS like ZScoreNormalization (same attribute names, other arithmetic) keeps its own __call__
    class Clipped(ZScore):
        def __call__(self, x):
            return np.clip(super().__call__(x), -2.0, 2.0)
    clipped = Clipped(np.full((1, 64), 3.0), np.full((1, 64), 2.0))
    ex3 = U.HighGammaExtractor(1000, 64, post_transforms=[clipped])
    assert not ex3._fused_post
    x3 = synthetic_ecog(79, 200, 64)
    assert np.array_equal(ex3.extract_features(x3), clipped(U.HighGammaExtractor(1000, 64).extract_features(x3)))


def test_vocoder_unit_segments_and_state_carry(oracle):
    import dss_amd.units as U
    from dss_amd import lpcnet
    blob = synthetic_blob(0)
    lpcnet.load_model(blob)
    voc = U.DelayedLPCNetVocoder()
    voc.initialize()
    dec = oracle.decoder(oracle.lpcnet_model(blob))
    for seed, L in ((11, 9), (12, 5)):                         # two segments through ONE decoder state
        seg = synthetic_features(seed, L).astype(np.float64)   # the unit casts to float32 itself (units.py:532)
        (_, msg), = asyncio.run(_drive(voc.synthesize(U.ClosedLoopMessage(data=seg, fs=100))))
        want = np.hstack([dec.synthesize(row) for row in seg.astype(np.float32)])
        assert msg.fs == 16000 and msg.data.dtype == np.int16 and np.array_equal(msg.data, want)
    voc.shutdown()


def test_decoder_model_on_gpu_matches_reference_golden(golden):
    """The BiLSTM the reference's RecurrentNeuralDecodingModel unit wraps (units.py:499-508): whole segment, fresh zero
    state, on PyTorch-ROCm; golden output from the reference's own local/models.py on CPU."""
    from dss_amd.models import BidirectionalSpeechSynthesisModel
    g = golden("models.npz")
    torch.manual_seed(0)
    m = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64).eval().cuda()
    with torch.no_grad():
        y, _ = m(torch.from_numpy(g["bilstm_in"]).cuda(), m.create_new_initial_state(batch_size=1, device="cuda"))
    np.testing.assert_allclose(y.cpu().numpy(), g["bilstm_out"], rtol=0, atol=2e-5)     # MIOpen LSTM vs CPU LSTM, fp32


def test_segment_pipeline_config3(oracle, golden):
    """64 segments of 1.04 s x 64 ch -> (64, 16000) PCM; every stage checked against its CPU counterpart."""
    from dss_amd import lpcnet
    from dss_amd.pipeline import SegmentPipeline
    blob = synthetic_blob(0)
    lpcnet.load_model(blob)
    B = 64
    ecog = np.stack([synthetic_ecog(1000 + b, 1040, 64) for b in range(B)])
    pipe = SegmentPipeline(B)
    pcm, hga, feats = pipe(torch.from_numpy(ecog).cuda(), return_intermediates=True)
    assert pcm.shape == (B, 16000) and hga.shape == (B, 100, 64) and feats.shape == (B, 100, 20)
    g = golden("hga_frames.npz")
    for b in range(4):                                         # HGA stage: device log, <= 1 ulp of the reference
        want = g[f"offline{b}_out"]
        assert (np.abs(hga[b].cpu().numpy() - want) / np.spacing(np.abs(want))).max() <= 1.0
    # vocoder stage: bit-exact against the oracle fed the SAME device-produced features
    m = oracle.lpcnet_model(blob)
    f0 = feats[5].cpu().numpy()
    assert np.array_equal(pcm[5].cpu().numpy(), oracle.lpcnet_utterance(m, f0))
    # segments are independent: a second call reproduces the first
    assert torch.equal(pipe(torch.from_numpy(ecog).cuda()), pcm)


def test_segment_pipeline_window_shape_beyond_the_lds_ring(oracle):
    """A window shape whose ring does not fit LDS (300 ms windows) sends the extractor to its three-launch form; the
    device-resident pipeline keeps the z-score inside the launch there too (hga_window_kernel's epilogue; this shape
    raised DSS_EINVAL on every call in round 3)."""
    from dss_amd import lpcnet
    from dss_amd.pipeline import SegmentPipeline
    lpcnet.load_model(synthetic_blob(0))
    B = 3
    rng = np.random.default_rng(2)
    mean, std = rng.standard_normal(64), rng.uniform(0.5, 2.0, 64)
    ecog = torch.from_numpy(np.stack([synthetic_ecog(300 + b, 1040, 64) for b in range(B)])).cuda()
    pipe = SegmentPipeline(B, channel_means=mean, channel_stds=std, window_length=0.3, window_shift=0.05)
    pcm, hga, feats = pipe(ecog, return_intermediates=True)          # frames before the z-score + host-side expression
    assert hga.shape[1] == pipe.frames and pcm.shape == (B, pipe.frames * 160)
    pcm2 = pipe(ecog)                                                # z-score as the launch's epilogue
    assert torch.equal(pcm, pcm2)
    small = SegmentPipeline(B, channel_means=mean, channel_stds=std)    # the reference's shape: the fused kernel, same contract
    a, _, _ = small(ecog, return_intermediates=True)
    assert torch.equal(a, small(ecog))


def test_streaming_pipeline_config5_parity(oracle, golden):
    """Config 5 at full size (128 streams, 40-sample packets) against the CPU side, tick by tick: stream 0 is fed the trial the
    reference-built golden frames were made from (hga_frames.npz::online_out, Cython + scipy in the same 40-sample chunking),
    so its HGA frames must equal them (device log: <= 1 ulp); the PCM of several streams must equal, bit for bit, one
    oracle decoder per stream fed the device-produced features of all ticks (decoder state carried across packets,
    units.py:524), through the eager first ticks and the graph-replayed steady state alike (decode_online.py:99-164)."""
    from dss_amd import lpcnet
    from dss_amd.pipeline import StreamingPipeline
    blob = synthetic_blob(0)
    lpcnet.load_model(blob)
    S, T = 128, 9
    g = golden("hga_frames.npz")
    x0 = synthetic_ecog(2000, 1040, 64)                        # the golden online trial
    rng = np.random.default_rng(11)
    sp = StreamingPipeline(S)
    check = (0, 1, 64, 127)
    feats = {s: [] for s in check}
    pcm = {s: [] for s in check}
    hga0 = []
    for k in range(T):
        pk = rng.standard_normal((S, 40, 64)) * 50.0
        pk[0] = x0[40 * k:40 * k + 40]
        out = sp.push(pk)
        W = sp.last_hga.shape[1]
        assert out.shape == (S, W * 160) and W == (1 if k == 0 else 4)
        hga0.append(sp.last_hga[0].cpu().numpy())
        f = sp.last_feats.cpu().numpy()
        for s in check:
            feats[s].append(f[s])
            pcm[s].append(out[s])
    assert sp._graph is not None                               # ticks 3.. were replayed from the captured graph
    got = np.concatenate(hga0)
    want = g["online_out"][:got.shape[0]]
    assert (np.abs(got - want) / np.spacing(np.abs(want))).max() <= 1.0
    m = oracle.lpcnet_model(blob)
    for s in check:
        assert np.array_equal(np.concatenate(pcm[s]), oracle.lpcnet_utterance(m, np.concatenate(feats[s]))), s


def test_streaming_pipeline_config5():
    from dss_amd import lpcnet
    from dss_amd.pipeline import StreamingPipeline
    lpcnet.load_model(synthetic_blob(0))
    S = 128
    sp = StreamingPipeline(S)
    rng = np.random.default_rng(0)
    first = sp.push(rng.standard_normal((S, 40, 64)) * 50)
    assert first.shape == (S, 160)                             # warm-start frame buffer: 1 frame from packet 1
    nxt = sp.push(rng.standard_normal((S, 40, 64)) * 50)
    assert nxt.shape == (S, 640) and nxt.dtype == np.int16     # then 4 frames = 40 ms of audio per packet
    lat = sp.measure_latency(20)
    assert np.isfinite(lat).all() and np.percentile(lat, 50) < 40.0   # must keep up with the 40 ms packet cadence
    # the steady-state tick replayed from a captured HIP graph gives the same PCM as eager launches
    a, b2 = StreamingPipeline(16, use_graph=True), StreamingPipeline(16, use_graph=False)
    rng = np.random.default_rng(5)
    for k in range(6):
        pk = rng.standard_normal((16, 40, 64)) * 50
        assert np.array_equal(a.push(pk), b2.push(pk)), k
    assert a._graph is not None or a._graph_failed


def test_asynchronous_synthesis_queue_file_contract(tmp_path, oracle):
    """.npy (N x 20) -> .wav 16 kHz, ragged lengths in one batch, bad files skipped."""
    from scipy.io.wavfile import read as wavread
    from dss_amd import lpcnet
    from dss_amd.synthesis_queue import AsynchronousSynthesisQueue
    blob = synthetic_blob(0)
    lpcnet.load_model(blob)
    m = oracle.lpcnet_model(blob)
    lengths = {"a": 7, "b": 4, "c": 11}
    for name, n in lengths.items():
        np.save(tmp_path / f"{name}.npy", synthetic_features(ord(name), n).astype(np.float64))
    np.save(tmp_path / "bad.npy", np.zeros((3, 5)))
    q = AsynchronousSynthesisQueue(nb_processes=8)
    for name in list(lengths) + ["bad", "missing"]:
        q.add_job(str(tmp_path / f"{name}.npy"))
    q.wait()
    for name, n in lengths.items():
        rate, pcm = wavread(tmp_path / f"{name}.wav")
        assert rate == 16000 and np.array_equal(pcm, oracle.lpcnet_utterance(m, synthetic_features(ord(name), n)))
    assert not (tmp_path / "bad.wav").exists()
    # a failing batched launch (here: made to fail) falls back to one launch per file, and an unwritable output path
    # costs only its own file (local/training.py:189-198 wraps load, synthesis and write per file)
    from dss_amd import lpcnet as lp
    for name in lengths:
        (tmp_path / f"{name}.wav").unlink()
    (tmp_path / "b.wav").mkdir()                       # wavwrite to a directory fails
    real = lp.LPCNetBatch.synthesize_ragged
    lp.LPCNetBatch.synthesize_ragged = lambda self, *a, **k: (_ for _ in ()).throw(RuntimeError("injected"))
    try:
        for name in lengths:
            q.add_job(str(tmp_path / f"{name}.npy"))
        q.wait()
    finally:
        lp.LPCNetBatch.synthesize_ragged = real
    for name in ("a", "c"):
        rate, pcm = wavread(tmp_path / f"{name}.wav")
        assert np.array_equal(pcm, oracle.lpcnet_utterance(m, synthetic_features(ord(name), lengths[name])))


def test_replay_of_a_recorded_session(tmp_path, oracle):
    """the logs decode_online.py leaves (raw packets, z-scored frames, decoded features) replayed through
    the GPU path.  The 'recorded' session is produced here by the CPU chain (oracle transforms + filters + frame buffer),
    written with the reference's logger format, and the replay must reproduce its frames bit for bit."""
    from ecog_chain_oracle import ZScore, reference_chain
    from dss_amd import formats, lpcnet, replay
    from dss_amd.hga import reference_filters
    lpcnet.load_model(synthetic_blob(0))
    both, car, speech = reference_chain()
    hg, fh, zi_hg, zi_fh = reference_filters(1000)
    ex = oracle.extractor({"sos_hg": hg, "sos_fh": fh, "zi_hg": zi_hg, "zi_fh": zi_fh}, 64)
    means, stds = np.full(64, 7.0), np.full(64, 1.5)
    z = ZScore(means, stds)
    raw = synthetic_ecog(4321, 40 * 12, 129)
    with open(tmp_path / "log.raw.f64", "wb") as fr, open(tmp_path / "log.hga.f64", "wb") as fh_:
        for pk in formats.iter_packets(raw, 40):
            formats.append_stream_log(fr, formats.parse_packet(formats.build_packet(pk)).astype(np.float64) * 0 + pk)
            formats.append_stream_log(fh_, z(ex.extract(speech(car(both(pk))))))
    feats = synthetic_features(77, 9)
    feats.tofile(tmp_path / "log.lpc.f32")
    np.save(tmp_path / "stats.npy", np.stack([means, stds]))
    rc = replay.main([str(tmp_path), "--normalization", str(tmp_path / "stats.npy"), "--wav", str(tmp_path / "out.wav")])
    assert rc == 0                                                   # frames bit-identical to the recorded ones
    from scipy.io.wavfile import read as wavread
    rate, pcm = wavread(tmp_path / "out.wav")
    m = oracle.lpcnet_model(synthetic_blob(0))
    assert rate == 16000 and np.array_equal(pcm, oracle.lpcnet_utterance(m, feats))
