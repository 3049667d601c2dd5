"""GPU tests of the speech-segment gate: csrc/speech_gate.hip through the C ABI against the
CPU restatement in oracle/speech_gate_oracle.py (pinned by hand-derived known answers in test_oracle_gate.py), and the
gated many-stream pipeline against the per-stream composition the reference's graph performs."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import pytest
import torch

from dss_amd.lpcnet_weights import synthetic_blob

pytestmark = pytest.mark.gpu


def _host_gate(C, N, ctx, sm_ctx):
    from speech_gate_oracle import SpeechGateOracle
    return SpeechGateOracle(C, N, ctx, sm_ctx)


@pytest.mark.parametrize("C,N,ctx,sm_ctx", [(64, 2000, 50, 5), (5, 37, 3, 2), (70, 16, 0, 0), (3, 23, 4, 1)])
def test_gate_matches_numpy_rings_bit_exact(C, N, ctx, sm_ctx):
    """Random label runs over many pushes of ragged size: same segments (float32 rows, bit for bit), same lengths, same
    speech counts -- including rings that wrap and speech runs longer than the ring (the classes' modulo arithmetic)."""
    from dss_amd.gate import SpeechGateGPU
    S, WMAX = 6, 7
    rng = np.random.default_rng(C * 1000 + N)
    gate = SpeechGateGPU(S, C, N, ctx, sm_ctx, 0.6, max_frames=WMAX)
    host = [_host_gate(C, N, ctx, sm_ctx) for _ in range(S)]
    state = rng.integers(0, 2, S)                       # label runs: flip with a per-stream probability
    p_flip = np.array([0.02, 0.05, 0.1, 0.2, 0.4, 0.01])
    n_seg = 0
    for tick in range(260):
        W = int(rng.integers(1, WMAX + 1))
        frames = rng.standard_normal((S, W, C)) * 3.0
        labels = np.zeros((S, W), dtype=np.int64)
        for i in range(W):
            state = np.where(rng.random(S) < p_flip, 1 - state, state)
            labels[:, i] = state
        got, n_speech = gate.push(frames, labels)
        for s in range(S):
            want, want_speech = host[s].push(frames[s], labels[s])
            assert len(got[s]) == len(want), (tick, s)
            assert n_speech[s] == want_speech
            for a, b in zip(got[s], want):
                assert a.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b), (tick, s)
            n_seg += len(want)
    assert n_seg > 20
    assert gate.frames_seen(0) > 0


def test_gate_against_the_reference_classes(golden):
    """tests/golden/gate.npz: what VoiceActivityDetectionSmoothing + SpeechSegmentHistory of the reference's own
    local/common.py returned (oracle/make_golden.py) -- the HIP gate replays every case bit for bit, each case on stream 1 of
    a 3-stream gate whose other streams see other data (a stream's rings are its own)."""
    from dss_amd.gate import SpeechGateGPU
    from test_oracle_gate import replay_gate_fixture
    g = golden("gate.npz")
    for ci in range(int(g["n_cases"][0])):
        C, N, ctx, sm = (int(v) for v in g[f"case{ci}_params"])
        gate = SpeechGateGPU(3, C, N, ctx, sm, 0.6, max_frames=8)
        rng = np.random.default_rng(ci)

        def push(frames, labels):
            W = len(labels)
            f = rng.standard_normal((3, W, C))
            lab = rng.integers(0, 2, (3, W))
            f[1], lab[1] = frames, labels
            segs, n_speech = gate.push(f, lab)
            return segs[1], int(n_speech[1])
        replay_gate_fixture(g, ci, push)


def test_gate_reset_one_stream_and_errors():
    from dss_amd import _lib
    from dss_amd.gate import SpeechGateGPU
    gate = SpeechGateGPU(2, 4, 32, 2, 1, 0.6, max_frames=6)
    f = np.ones((2, 6, 4))
    lab = np.array([[1, 1, 1, 1, 0, 0]] * 2)
    gate.push(f, lab)
    gate.reset(1)
    assert gate.frames_seen(0) == 6 and gate.frames_seen(1) == 0
    segs, _ = gate.push(np.ones((2, 3, 4)), np.zeros((2, 3)))
    assert len(segs[0]) == 1 and len(segs[1]) == 0          # stream 0 closes its run, stream 1 forgot it
    with pytest.raises(_lib.DssError, match="frames per push"):
        gate.push(np.ones((2, 7, 4)), np.zeros((2, 7)))
    with pytest.raises(_lib.DssError, match="<= 64"):
        SpeechGateGPU(1, 4, 32, 2, 40)


class _ThresholdVAD(torch.nn.Module):
    """Stands in for a trained VAD checkpoint (none exists offline): speech when the first z-scored feature is positive.
    Same call surface as dss_amd.models.UnidirectionalVoiceActivityDetector."""

    def create_new_initial_state(self, batch_size, device="cpu", req_grad=False):
        return (torch.zeros(1, batch_size, 1, device=device), torch.zeros(1, batch_size, 1, device=device))

    def forward(self, x, state=None):
        return torch.stack([torch.zeros_like(x[..., 0]), x[..., 0]], dim=-1), state


def test_gated_streaming_pipeline_against_per_stream_composition(oracle):
    """decode_online.py's chain for several streams at once: segments, previous_frames and PCM per stream equal what the
    per-stream gate oracle (reference common.py:106-215 + units.py:432-447) and the vocoder oracle produce from the same
    frames, with the vocoder state of a stream carried from one of its segments to the next."""
    from dss_amd import lpcnet
    from dss_amd.pipeline import GatedStreamingPipeline
    blob = synthetic_blob(0)
    lpcnet.load_model(blob)
    model = oracle.lpcnet_model(blob)
    S, C, ticks = 3, 64, 70
    rng = np.random.default_rng(7)
    # ECoG whose amplitude alternates between quiet and loud stretches, so that log power crosses the z-score mean
    env = np.ones((S, ticks * 40))
    for s in range(S):
        t = 0
        loud = False
        while t < env.shape[1]:
            n = int(rng.integers(150, 500))
            env[s, t:t + n] = 400.0 if loud else 20.0
            loud = not loud
            t += n
    ecog = rng.standard_normal((S, ticks * 40, C)) * env[:, :, None]
    mean = np.full(C, 7.4)                              # between the two stretches' log band power (about 4.4 and 10.4)
    pipe = GatedStreamingPipeline(S, C, buffer_size=300, context_frames=8, channel_means=mean, vad=_ThresholdVAD(),
                                  max_segment_frames=300, asynchronous=False)      # the reference's own behaviour: the tick waits
    host = [_host_gate(C, 300, 8, 5) for _ in range(S)]
    vocoders = [oracle.decoder(model) for _ in range(S)]
    counter, n_seg = 0, 0
    for k in range(ticks):
        got = pipe.push(ecog[:, k * 40:(k + 1) * 40])
        z, labels = pipe.last_z.cpu().numpy(), pipe.last_labels.cpu().numpy()
        W = z.shape[1]
        counter += W
        want = []
        for s in range(S):
            segs, n_sp = host[s].push(z[s], labels[s])
            for seg in segs:
                x = torch.from_numpy(seg)[None].cuda()
                y = pipe._decode(x)                      # the decoder as the pipeline runs it (csrc/bilstm_decoder.hip) ...
                with torch.no_grad():                    # ... which agrees with the PyTorch-ROCm module on the same weights
                    yt, _ = pipe.decoder(x, pipe.decoder.create_new_initial_state(batch_size=1, device="cuda"))
                assert pipe.dec_gpu is not None and (y - yt).abs().max().item() <= 2e-5
                feats = y[0].cpu().numpy()
                pcm = np.concatenate([vocoders[s].synthesize(feats[t]) for t in range(len(feats))])
                want.append((s, counter - len(seg) - (W - n_sp), pcm))
        assert [(s, p) for s, p, _ in got] == [(s, p) for s, p, _ in want], k
        for (_, _, a), (_, _, b) in zip(got, want):
            assert a.dtype == np.int16 and np.array_equal(a, b), k
        n_seg += len(want)
    assert n_seg >= 6


def test_vad_kernel_matches_the_reference_golden_and_torch(golden):
    """Row f4's LSTM(150) x 2 step kernel (csrc/vad_lstm.hip) against the reference's own detector: the golden logits that
    /root/reference's UnidirectionalVoiceActivityDetector produced (oracle/make_golden.py: torch.manual_seed(1), two packets
    of four frames, state carried) within 2e-5, and on random frames for 128 streams the labels (and logits, 2e-5) of
    torch.nn.LSTM running the same weights on the GPU, over several packets with carried state."""
    from dss_amd.models import UnidirectionalVoiceActivityDetector
    from dss_amd.vad import VadLstmGPU, fits
    g = golden("models.npz")
    torch.manual_seed(1)                                   # the seed the golden vector's weights were drawn with
    m = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=64).eval()
    assert fits(m) and sum(p.numel() for p in m.parameters()) == int(g["vad_params"][0])
    x = torch.from_numpy(g["bilstm_in"]).cuda()            # (1, 100, 64) float32: the golden run used its first 8 frames
    k = VadLstmGPU(1, m)
    l1, y1 = k.step_torch(x[:, :4], want_logits=True)
    l2, y2 = k.step_torch(x[:, 4:8].to(torch.float64), want_logits=True)      # the float64 entry casts like units.py:433
    got = torch.cat([y1, y2], dim=1).cpu().numpy()
    assert np.abs(got - g["vad_out"]).max() <= 2e-5
    assert np.array_equal(torch.cat([l1, l2], dim=1).cpu().numpy(), g["vad_out"].argmax(axis=2).astype(np.int32))
    # 128 streams x packets of 4, 1, 5 and 4 frames against torch.nn.LSTM on the same device, state carried by both
    S = 128
    mg = m.cuda()
    k = VadLstmGPU(S, mg)
    state = mg.create_new_initial_state(batch_size=S, device="cuda")
    rng = np.random.default_rng(9)
    n_close = 0
    for w in (4, 1, 5, 4, 4):
        z = torch.from_numpy(rng.standard_normal((S, w, 64)) * 2.0).cuda()
        with torch.no_grad():
            want, state = mg(z.to(torch.float32), state)
        labels, logits = k.step_torch(z, want_logits=True)
        assert (logits - want).abs().max().item() <= 2e-5
        margin = (want[..., 1] - want[..., 0]).abs()
        sure = margin > 1e-4                               # a tie within the tolerance may fall either way
        n_close += int((~sure).sum())
        assert torch.equal(labels[sure], want.argmax(dim=2).to(torch.int32)[sure])
    assert n_close < 20
    h, c = k.state()
    assert np.abs(h - state[0].cpu().numpy()).max() <= 2e-5 and np.abs(c - state[1].cpu().numpy()).max() <= 1e-4
    # a stream's state can be cleared on its own
    k.reset(3)
    h2, c2 = k.state()
    assert not h2[:, 3].any() and not c2[:, 3].any() and np.array_equal(h2[:, 4], h[:, 4])


def test_gated_pipeline_with_the_vad_kernel_equals_the_torch_detector():
    """GatedStreamingPipeline with the reference's detector: the kernel path (default) and the PyTorch-ROCm module give the
    same labels and therefore the same segments and PCM."""
    from dss_amd import lpcnet
    from dss_amd.pipeline import GatedStreamingPipeline
    lpcnet.load_model(synthetic_blob(0))
    from dss_amd.models import UnidirectionalVoiceActivityDetector
    S, C = 6, 64
    rng = np.random.default_rng(21)
    # stretches of loud and quiet input and a detector whose two logits are mirror images (w1 = -w0, no bias), so that its
    # decision follows the sign of one projection of the LSTM state and both labels occur
    env = np.repeat(np.where(rng.random((S, 60)) < 0.5, 3.0, 60.0), 20, axis=1)          # (S, 1200)
    ecog = rng.standard_normal((S, 1200, C)) * env[:, :, None]
    pk = [ecog[:, 40 * k:40 * k + 40] for k in range(30)]
    mean = np.full(C, 5.0)

    def detector():
        torch.manual_seed(5)
        m = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=C)
        with torch.no_grad():
            m.classifier.weight[1] = -m.classifier.weight[0]
            m.classifier.bias.zero_()
        return m
    kw = dict(buffer_size=300, context_frames=8, max_segment_frames=300, channel_means=mean, asynchronous=False)
    a = GatedStreamingPipeline(S, C, vad=detector(), **kw)
    b = GatedStreamingPipeline(S, C, vad=detector(), use_vad_kernel=False, **kw)
    assert a.vad_gpu is not None and b.vad_gpu is None
    n_labels = 0
    for p in pk:
        ga, gb = a.push(p), b.push(p)
        la, lb = a.last_labels.cpu().numpy(), b.last_labels.cpu().numpy()
        assert np.array_equal(la, lb)                     # (a logit tie within 1e-6 could differ; none occurs with this seed)
        n_labels += int(la.sum())
        assert [(s, q) for s, q, _ in ga] == [(s, q) for s, q, _ in gb]
        for (_, _, x), (_, _, y) in zip(ga, gb):
            assert np.array_equal(x, y)
    assert 0 < n_labels < S * 30 * 4                      # the seeded detector says both things


def _loud_quiet(rng, S, n, lo, hi, a=150, b=500):
    env = np.ones((S, n))
    for s in range(S):
        t, loud = 0, bool(rng.integers(2))
        while t < n:
            k = int(rng.integers(a, b))
            env[s, t:t + k] = hi if loud else lo
            loud, t = not loud, t + k
    return env


def test_asynchronous_segment_synthesis_equals_the_blocking_path():
    """The many-stream gated mode with vocoding OFF the tick path (SegmentSynthesisQueue: closing segments collected into a
    pool, ragged decoder + ragged vocoder launch on side streams, PCM by event) against the blocking path (asynchronous=False,
    what the reference's single stream does, units.py:531-538): per stream the same segments, in the same order, with the
    same previous_frames and bit-identical PCM -- a stream's vocoder state carries from segment to segment in both.  Runs
    with fewer rows per job than streams and a single lane as well, so that segments queue behind each other."""
    from dss_amd import lpcnet
    from dss_amd.pipeline import GatedStreamingPipeline
    lpcnet.load_model(synthetic_blob(0))
    S, C, ticks = 12, 64, 90
    rng = np.random.default_rng(77)
    env = _loud_quiet(rng, S, ticks * 40, 20.0, 400.0, 120, 420)
    ecog = rng.standard_normal((S, ticks * 40, C)) * env[:, :, None]
    kw = dict(buffer_size=300, context_frames=8, channel_means=np.full(C, 7.4), vad=_ThresholdVAD(), max_segment_frames=300)
    ref = GatedStreamingPipeline(S, C, asynchronous=False, **kw)
    want = []
    for k in range(ticks):
        want += ref.push(ecog[:, k * 40:(k + 1) * 40])
    assert len(want) >= 20 and ref.flush() == []
    for lanes, rows, pool in ((3, 32, None), (1, 2, None), (2, 1, 3)):     # the last one: a pool of three segment buffers -- submit() has to wait for rows
        pipe = GatedStreamingPipeline(S, C, asynchronous=True, n_lanes=lanes, rows_per_job=rows, pool_rows=pool, **kw)
        got, seen_pending = [], 0
        for k in range(ticks):
            got += pipe.push(ecog[:, k * 40:(k + 1) * 40])
            seen_pending = max(seen_pending, pipe.queue.in_flight)
        got += pipe.flush()
        assert pipe.queue.in_flight == 0 and pipe.segments_closed == len(want) == len(got)
        assert seen_pending > 0                               # ticks really returned while segments were still in flight
        for s in range(S):                                    # per stream: same sequence (closing order), same audio
            a = [(p, pcm) for st, p, pcm in got if st == s]
            b = [(p, pcm) for st, p, pcm in want if st == s]
            assert [p for p, _ in a] == [p for p, _ in b], (lanes, rows, s)
            for (_, x), (_, y) in zip(a, b):
                assert x.dtype == np.int16 and np.array_equal(x, y), (lanes, rows, s)
        assert len(pipe.queue.latencies_ms) == len(got)
        pipe.close()
        assert pipe.queue._thread is None and pipe.queue.lanes == []
        del pipe


def test_gated_pipeline_from_wire_format_packets():
    """push_wire (packet bodies as they arrive: float32, channel-major) against push on the packets parsed on the host: the same
    labels, frames, segments and PCM."""
    from dss_amd import formats as F, lpcnet
    from dss_amd.pipeline import GatedStreamingPipeline
    lpcnet.load_model(synthetic_blob(0))
    S, C, ticks = 3, 64, 50
    rng = np.random.default_rng(31)
    env = _loud_quiet(rng, S, ticks * 40, 20.0, 400.0)
    kw = dict(buffer_size=300, context_frames=8, channel_means=np.full(C, 7.4), vad=_ThresholdVAD(), max_segment_frames=300, asynchronous=False)
    a, b = GatedStreamingPipeline(S, C, **kw), GatedStreamingPipeline(S, C, **kw)
    n_seg = 0
    for k in range(ticks):
        x = rng.standard_normal((S, 40, C)) * env[:, k * 40:(k + 1) * 40, None]
        packets = [F.build_packet(x[s]) for s in range(S)]
        ga = a.push(np.stack([F.parse_packet(p) for p in packets]))
        gb = b.push_wire(np.stack([F.packet_payload(p) for p in packets]))
        assert torch.equal(a.last_z, b.last_z) and torch.equal(a.last_labels, b.last_labels)
        assert [(s, q) for s, q, _ in ga] == [(s, q) for s, q, _ in gb]
        for (_, _, u), (_, _, v) in zip(ga, gb):
            assert np.array_equal(u, v)
        n_seg += len(ga)
    assert n_seg >= 3
    a.close(); b.close()


def test_segment_queue_refuses_and_reports():
    """A segment longer than the queue's buffers is refused on the tick that closes it (nothing is truncated); a failure on the
    worker thread surfaces on the caller's next call instead of vanishing with the thread."""
    from dss_amd import lpcnet
    from dss_amd.pipeline import GatedStreamingPipeline
    lpcnet.load_model(synthetic_blob(0))
    S, C = 2, 64
    rng = np.random.default_rng(5)
    kw = dict(buffer_size=300, context_frames=8, channel_means=np.full(C, 7.4), vad=_ThresholdVAD())
    pipe = GatedStreamingPipeline(S, C, max_segment_frames=20, **kw)                  # speech runs of ~60 frames + 16 of context
    env = np.concatenate([np.full(200, 20.0), np.full(600, 400.0), np.full(800, 20.0)])
    ecog = rng.standard_normal((S, env.size, C)) * env[None, :, None]
    with pytest.raises(ValueError, match="exceeds max_segment_frames"):
        for k in range(env.size // 40):
            pipe.push(ecog[:, k * 40:(k + 1) * 40])
    pipe.close()
    pipe = GatedStreamingPipeline(S, C, max_segment_frames=300, **kw)

    def boom(lane, job):
        raise RuntimeError("injected")
    pipe.queue._launch = boom
    with pytest.raises(RuntimeError, match="worker failed"):
        for k in range(env.size // 40):
            pipe.push(ecog[:, k * 40:(k + 1) * 40])
        pipe.flush()
    pipe.close()


def test_asynchronous_path_at_full_size_128_streams():
    """BASELINE config 5 as decode_online.py runs it, at its own sizes: 128 streams, ring 2000, context 50, the reference's detector and
    decoder architectures on the library's kernels, seven lanes when the hardware queues allow.  Every segment that closes comes back
    exactly once, per stream in closing order, with the blocking path's previous_frames and bit-identical PCM."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gated_leg
    from dss_amd import lpcnet
    from dss_amd.pipeline import GatedStreamingPipeline
    lpcnet.load_model(synthetic_blob(0))
    S, ticks = 128, 130
    packets = gated_leg.make_input()[:ticks]
    kw = dict(channel_means=np.full(64, 5.0), max_segment_frames=600)
    ref = GatedStreamingPipeline(S, 64, vad=gated_leg.detector(), asynchronous=False, **kw)
    pipe = GatedStreamingPipeline(S, 64, vad=gated_leg.detector(), **kw)
    assert pipe.vad_gpu is not None and pipe.dec_gpu is not None and len(pipe.queue.lanes) >= 3
    want, got = [], []
    for k in range(ticks):
        want += ref.push(packets[k])
        got += pipe.push(packets[k])
    got += pipe.flush()
    assert len(want) >= 30 and len(got) == len(want) == pipe.segments_closed
    by = {}
    for s, p, pcm in want:
        by.setdefault(s, []).append((p, pcm))
    seen = {}
    for s, p, pcm in got:
        k = seen.get(s, 0)
        assert k < len(by[s]) and by[s][k][0] == p and np.array_equal(by[s][k][1], pcm), (s, k)
        seen[s] = k + 1
    assert all(seen.get(s, 0) == len(v) for s, v in by.items())
    ref.close(); pipe.close()


def test_asynchronous_queue_with_a_module_of_another_architecture():
    """A decoder the kernels do not take (3 layers) runs as the PyTorch-ROCm module on the lane's stream: same PCM as the
    blocking path."""
    from dss_amd import lpcnet
    from dss_amd.models import BidirectionalSpeechSynthesisModel
    from dss_amd.pipeline import GatedStreamingPipeline
    lpcnet.load_model(synthetic_blob(0))
    S, C, ticks = 4, 64, 60
    rng = np.random.default_rng(78)
    ecog = rng.standard_normal((S, ticks * 40, C)) * _loud_quiet(rng, S, ticks * 40, 20.0, 400.0)[:, :, None]

    def model():
        torch.manual_seed(3)
        return BidirectionalSpeechSynthesisModel(nb_layer=3, nb_hidden_units=24, nb_electrodes=C)
    kw = dict(buffer_size=300, context_frames=8, channel_means=np.full(C, 7.4), vad=_ThresholdVAD(), max_segment_frames=300)
    a = GatedStreamingPipeline(S, C, decoder=model(), asynchronous=False, **kw)
    b = GatedStreamingPipeline(S, C, decoder=model(), asynchronous=True, **kw)
    assert a.dec_gpu is None and b.dec_gpu is None
    want, got = [], []
    for k in range(ticks):
        want += a.push(ecog[:, k * 40:(k + 1) * 40])
        got += b.push(ecog[:, k * 40:(k + 1) * 40])
    got += b.flush()
    assert len(want) >= 4 and sorted((s, p) for s, p, _ in got) == sorted((s, p) for s, p, _ in want)
    d = {(s, p): pcm for s, p, pcm in want}
    for s, p, pcm in got:
        assert np.array_equal(pcm, d[(s, p)])


def test_look_alike_modules_keep_their_own_forward():
    """fits() reads parameter names and shapes; make_kernel() also runs the module: a decoder / detector with the reference's
    parameters but another forward (a clamp behind the head, float64 weights) is NOT swapped for the kernels."""
    from dss_amd import decoder as D, vad as V
    from dss_amd.models import BidirectionalSpeechSynthesisModel, UnidirectionalVoiceActivityDetector

    class Clamped(BidirectionalSpeechSynthesisModel):
        def forward(self, x, state=None):
            y, st = super().forward(x, state)
            return torch.clamp(y, -0.01, 0.01), st

    class Scaled(UnidirectionalVoiceActivityDetector):
        def forward(self, x, state=None):
            return super().forward(x * 3.0, state)
    torch.manual_seed(0)
    plain = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64).eval().cuda()
    assert D.make_kernel(plain, 1, 16) is not None
    look = Clamped(nb_layer=2, nb_hidden_units=100, nb_electrodes=64).eval().cuda()
    assert D.fits(look)
    with pytest.warns(RuntimeWarning, match="forward differs"):
        assert D.make_kernel(look, 1, 16) is None
    assert not D.fits(BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64).double())
    det = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=64).eval().cuda()
    assert V.make_kernel(det, 4) is not None
    look = Scaled(nb_layer=2, nb_hidden_units=150, nb_electrodes=64).eval().cuda()
    assert V.fits(look)
    with pytest.warns(RuntimeWarning, match="forward differs"):
        assert V.make_kernel(look, 4) is None
    assert not V.fits(UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=64).double())


@pytest.mark.parametrize("S,W", [(257, 4), (513, 5)])
def test_vad_kernel_two_streams_per_workgroup(S, W):
    """Beyond 256 streams a workgroup steps two streams (the SW = 2 instantiation; an odd count leaves the last one half
    empty): logits against torch.nn.LSTM over several packets with carried state, and reset of one stream."""
    from dss_amd.models import UnidirectionalVoiceActivityDetector
    from dss_amd.vad import VadLstmGPU
    torch.manual_seed(11)
    m = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=64).eval().cuda()
    k = VadLstmGPU(S, m)
    state = m.create_new_initial_state(batch_size=S, device="cuda")
    rng = np.random.default_rng(S)
    for w in (W, 1, W):
        z = torch.from_numpy(rng.standard_normal((S, w, 64)) * 2.0).cuda()
        with torch.no_grad():
            want, state = m(z.to(torch.float32), state)
        labels, logits = k.step_torch(z, want_logits=True)
        assert (logits - want).abs().max().item() <= 2e-5
        sure = (want[..., 1] - want[..., 0]).abs() > 1e-4
        assert torch.equal(labels[sure], want.argmax(dim=2).to(torch.int32)[sure])
    h, c = k.state()
    assert np.abs(h - state[0].cpu().numpy()).max() <= 2e-5 and np.abs(c - state[1].cpu().numpy()).max() <= 1e-4
    k.reset(S - 1)                                          # the odd stream out, on the steps' stream
    k.reset(2)
    h2, c2 = k.state()
    assert not h2[:, S - 1].any() and not c2[:, 2].any() and np.array_equal(h2[:, 3], h[:, 3])
    state = (state[0].clone(), state[1].clone())
    state[0][:, S - 1] = 0; state[1][:, S - 1] = 0; state[0][:, 2] = 0; state[1][:, 2] = 0
    z = torch.from_numpy(rng.standard_normal((S, 3, 64))).cuda()
    with torch.no_grad():
        want, state = m(z.to(torch.float32), state)
    _, logits = k.step_torch(z, want_logits=True)
    assert (logits - want).abs().max().item() <= 2e-5


@pytest.mark.parametrize("S,C,H", [(1, 5, 7), (3, 64, 150), (5, 17, 33), (2, 128, 160)])
def test_vad_kernel_odd_shapes(S, C, H):
    """Input and hidden sizes that are not multiples of 4 (the kernel's weight copies are padded), stream counts that do not
    fill the last workgroup, the largest sizes the kernel takes: logits and state against torch.nn.LSTM, float32 and float64
    frames; sizes beyond the limits are refused, not truncated."""
    from dss_amd.models import UnidirectionalVoiceActivityDetector
    from dss_amd.vad import VadLstmGPU, fits
    torch.manual_seed(100 + H)
    m = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=H, nb_electrodes=C).eval().cuda()
    assert fits(m)
    k = VadLstmGPU(S, m)
    state = m.create_new_initial_state(batch_size=S, device="cuda")
    rng = np.random.default_rng(H)
    for w, dt in ((3, torch.float32), (1, torch.float64), (6, torch.float64)):
        z = torch.from_numpy(rng.standard_normal((S, w, C))).cuda().to(dt)
        with torch.no_grad():
            want, state = m(z.to(torch.float32), state)
        labels, logits = k.step_torch(z, want_logits=True)
        assert (logits - want).abs().max().item() <= 2e-5
        sure = (want[..., 1] - want[..., 0]).abs() > 1e-4
        assert torch.equal(labels[sure], want.argmax(dim=2).to(torch.int32)[sure])
    h, c = k.state()
    assert np.abs(h - state[0].cpu().numpy()).max() <= 2e-5 and np.abs(c - state[1].cpu().numpy()).max() <= 1e-4
    big = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=512, nb_electrodes=128)      # the class's own defaults
    assert not fits(big)                                   # beyond the kernel: the pipeline keeps such a model on PyTorch-ROCm
    three = UnidirectionalVoiceActivityDetector(nb_layer=3, nb_hidden_units=32, nb_electrodes=8)
    assert not fits(three)
