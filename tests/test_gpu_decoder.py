"""GPU tests of the bidirectional decoder kernels (csrc/bilstm_decoder.hip) through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(C=64, H=100, seed=0):
    from dss_amd.models import BidirectionalSpeechSynthesisModel
    torch.manual_seed(seed)
    return BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=H, nb_electrodes=C).eval()


def test_decoder_kernel_matches_the_reference_golden_and_torch(golden):
    """The three-launch BiLSTM against the reference's own decoder: the golden features /root/reference's
    BidirectionalSpeechSynthesisModel produced on the CPU (oracle/make_golden.py: torch.manual_seed(0), one segment of 100
    frames, zero state) within 2e-5 -- the tolerance the PyTorch-ROCm module itself is tested at -- for float32 and float64
    frames; and, for 128 streams x 4 frames (the streaming tick) and 64 x 100 (config 3), torch.nn.LSTM running the same
    weights on the GPU."""
    from dss_amd.decoder import BiLstmDecoderGPU, fits
    g = golden("models.npz")
    m = _model()
    assert fits(m) and sum(p.numel() for p in m.parameters()) == int(g["bilstm_params"][0])
    x = torch.from_numpy(g["bilstm_in"]).cuda()            # (1, 100, 64) float32
    k = BiLstmDecoderGPU(4, 128, m)
    y32 = k(x).cpu().numpy()
    y64 = k(x.to(torch.float64)).cpu().numpy()            # the float64 entry casts like units.py:503
    assert y32.shape == (1, 100, 20) and np.array_equal(y32, y64)
    assert np.abs(y32 - g["bilstm_out"]).max() <= 2e-5
    mg = m.cuda()
    rng = np.random.default_rng(4)
    for S, T in ((128, 4), (64, 100), (5, 1)):
        k = BiLstmDecoderGPU(S, T, mg)
        z = torch.from_numpy(rng.standard_normal((S, T, 64)) * 2.0).cuda()
        with torch.no_grad():
            want, _ = mg(z.to(torch.float32), mg.create_new_initial_state(batch_size=S, device="cuda"))
        got = k(z)
        assert got.shape == want.shape and (got - want).abs().max().item() <= 2e-5
        # every call starts from the zero state: the same input gives the same output again
        assert torch.equal(k(z), got)
        # fewer streams / frames than the handle was built for
        if S > 2 and T > 2:
            part = k(z[:S - 1, :T - 1])
            with torch.no_grad():
                wp, _ = mg(z[:S - 1, :T - 1].to(torch.float32), mg.create_new_initial_state(batch_size=S - 1, device="cuda"))
            assert (part - wp).abs().max().item() <= 2e-5


@pytest.mark.parametrize("S,T,C,H", [(1, 3, 5, 7), (3, 6, 64, 100), (6, 2, 17, 33), (2, 5, 256, 128)])
def test_decoder_kernel_odd_shapes(S, T, C, H):
    """Input and hidden sizes that are not multiples of 4 (the kernel's weight copies are padded), stream counts that do not
    fill the last workgroup, the largest sizes the kernel takes; sizes and architectures beyond it are refused, not truncated."""
    from dss_amd.decoder import BiLstmDecoderGPU, fits
    from dss_amd.models import BidirectionalSpeechSynthesisModel, UnidirectionalVoiceActivityDetector
    m = _model(C, H, seed=200 + H).cuda()
    assert fits(m)
    k = BiLstmDecoderGPU(S, T, m)
    rng = np.random.default_rng(H)
    for dt in (torch.float32, torch.float64):
        z = torch.from_numpy(rng.standard_normal((S, T, C))).cuda().to(dt)
        with torch.no_grad():
            want, _ = m(z.to(torch.float32), m.create_new_initial_state(batch_size=S, device="cuda"))
        assert (k(z) - want).abs().max().item() <= 2e-5
    with pytest.raises(ValueError):
        k(torch.zeros((S + 1, T, C), device="cuda"))
    with pytest.raises(ValueError):
        k(torch.zeros((S, T + 1, C), device="cuda"))
    assert not fits(BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=256, nb_electrodes=64))
    assert not fits(BidirectionalSpeechSynthesisModel(nb_layer=3, nb_hidden_units=32, nb_electrodes=8))
    assert not fits(UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=32, nb_electrodes=8))


def test_pipelines_with_the_decoder_kernel_agree_with_the_torch_decoder():
    """SegmentPipeline and StreamingPipeline with the reference's decoder: the kernel path (default) and the PyTorch-ROCm
    module give features within 2e-5 of each other on the same input, and each path's PCM is what the vocoder makes of ITS
    features (the vocoder is bit-exact per feature vector, tests/test_gpu_units.py)."""
    from dss_amd.lpcnet import load_model
    from dss_amd.lpcnet_weights import synthetic_blob
    from dss_amd.pipeline import SegmentPipeline, StreamingPipeline
    from dss_amd.synthetic import synthetic_ecog
    load_model(synthetic_blob(0))
    B = 4
    x = torch.from_numpy(np.stack([synthetic_ecog(70 + b, 1040, 64) for b in range(B)])).cuda()
    a = SegmentPipeline(B, 1040, 64, channel_means=np.full(64, 4.0), channel_stds=np.full(64, 1.5), seed=2)
    b = SegmentPipeline(B, 1040, 64, channel_means=np.full(64, 4.0), channel_stds=np.full(64, 1.5), seed=2, use_decoder_kernel=False)
    assert a.dec_gpu is not None and b.dec_gpu is None
    pa, _, fa = a(x, return_intermediates=True)
    pb, _, fb = b(x, return_intermediates=True)
    assert fa.shape == fb.shape and (fa - fb).abs().max().item() <= 2e-5
    assert pa.shape == pb.shape and pa.dtype == torch.int16
    # the streaming tick, eager and graph-replayed, against the module on the same frames
    S = 8
    sk = StreamingPipeline(S, 64, seed=2)
    sm = StreamingPipeline(S, 64, seed=2, use_decoder_kernel=False)
    assert sk.dec_gpu is not None and sm.dec_gpu is None
    rng = np.random.default_rng(3)
    for _ in range(5):
        pk = rng.standard_normal((S, 40, 64)) * 50.0
        ya, yb = sk.push(pk), sm.push(pk)
        assert ya.shape == yb.shape
        assert torch.equal(sk.last_hga, sm.last_hga)
        assert (sk.last_feats - sm.last_feats).abs().max().item() <= 2e-5


def test_decoder_unit_runs_the_kernels_for_the_reference_architecture(golden, tmp_path):
    """dss_amd.units.RecurrentNeuralDecodingModel (the surface of units.py:450-508) on the GPU box: the reference's
    architecture takes the kernel path (golden features within 2e-5, the PyTorch-ROCm module within 2e-5, a fresh state per
    segment, segments of different lengths), a model of another architecture stays on the module."""
    import asyncio
    import dss_amd.units as U
    from dss_amd.models import BidirectionalSpeechSynthesisModel
    g = golden("models.npz")
    ref = _model()
    path = tmp_path / "decoder.pth"
    torch.save(ref.state_dict(), path)
    unit = U.RecurrentNeuralDecodingModel(U.RecurrentNeuralDecodingModelSettings(
        path_to_model_weights=str(path), model=BidirectionalSpeechSynthesisModel,
        params=dict(nb_layer=2, nb_hidden_units=100, nb_electrodes=64)))
    unit.initialize()
    assert unit.STATE.kernel is not None and unit.STATE.device == "cuda"

    async def drive(gen):
        return [item async for item in gen]
    x = g["bilstm_in"][0]
    mg = ref.cuda()
    for L in (100, 37, 1, 100):
        (stream, msg), = asyncio.run(drive(unit.decode(U.ClosedLoopMessage(data=x[:L].astype(np.float64), fs=100))))
        assert stream is unit.OUTPUT and msg.fs == 100 and msg.data.shape == (L, 20) and msg.data.dtype == np.float32
        with torch.no_grad():
            want, _ = mg(torch.from_numpy(x[:L])[None].cuda(), mg.create_new_initial_state(batch_size=1, device="cuda"))
        assert np.abs(msg.data - want[0].cpu().numpy()).max() <= 2e-5
        if L == 100:
            assert np.abs(msg.data - g["bilstm_out"][0]).max() <= 2e-5
    other = U.RecurrentNeuralDecodingModel(U.RecurrentNeuralDecodingModelSettings(
        path_to_model_weights=None, model=BidirectionalSpeechSynthesisModel,
        params=dict(nb_layer=3, nb_hidden_units=24, nb_electrodes=64)))
    other.initialize()
    assert other.STATE.kernel is None
    (_, msg), = asyncio.run(drive(other.decode(U.ClosedLoopMessage(data=x[:9].astype(np.float64), fs=100))))
    assert msg.data.shape == (9, 20)


@pytest.mark.parametrize("S,T", [(129, 4), (257, 5), (1024, 4), (1023, 5)])
def test_decoder_kernel_several_streams_per_workgroup(S, T):
    """Beyond 128 streams a workgroup carries two streams (129 .. 256) or four (the W = 2 / W = 4 instantiations, [unit][stream]
    LDS layout, cell ownership by (stream, unit)); odd stream counts leave the last workgroup partly empty.  Against
    torch.nn.LSTM on the same weights, 2e-5."""
    from dss_amd.decoder import BiLstmDecoderGPU
    m = _model().cuda()
    k = BiLstmDecoderGPU(S, T, m)
    z = torch.from_numpy(np.random.default_rng(S + T).standard_normal((S, T, 64)) * 2.0).cuda()
    with torch.no_grad():
        want, _ = m(z.to(torch.float32), m.create_new_initial_state(batch_size=S, device="cuda"))
    got = k(z)
    assert got.shape == want.shape and (got - want).abs().max().item() <= 2e-5
    assert torch.equal(k(z), got)
    f32 = k(z.to(torch.float32))
    assert torch.equal(f32, got)


@pytest.mark.parametrize("n", [5, 150, 300])
def test_decoder_ragged_rows_equal_one_call_per_segment(n):
    """dss_dec_forward_rows_dev: n segments of different lengths (one of them empty), read from scattered rows of a pool whose
    rows are longer than the call, in one call -- each segment's features are bit-identical to the plain call on that segment
    alone (same kernels, same order of operations; the backward direction starts at the segment's own last frame), nothing is
    written beyond a segment's length.  n = 150 / 300 run two / four segments per workgroup with unequal lengths."""
    from dss_amd.decoder import BiLstmDecoderGPU
    m = _model().cuda()
    rng = np.random.default_rng(n)
    cap, fmax = 40, 33
    counts = rng.integers(1, fmax + 1, n)
    counts[1] = 0
    counts[2] = fmax
    rows = rng.permutation(n + 7)[:n]
    pool = torch.from_numpy(rng.standard_normal((n + 7, cap, 64)).astype(np.float32) * 2.0).cuda()
    k = BiLstmDecoderGPU(n, fmax, m)
    feats = torch.full((n, fmax, 20), 777.0, dtype=torch.float32, device="cuda")
    k.forward_rows_torch(pool, rows, counts, feats, fmax)
    one = BiLstmDecoderGPU(1, fmax, m)
    for i in (list(range(n)) if n <= 5 else list(range(0, n, max(1, n // 24))) + [n - 1]):
        L = int(counts[i])
        if L:
            want = one(pool[int(rows[i]), :L][None])[0]
            assert torch.equal(feats[i, :L], want), i
            with torch.no_grad():
                ref, _ = m(pool[int(rows[i]), :L][None], m.create_new_initial_state(batch_size=1, device="cuda"))
            assert (feats[i, :L] - ref[0]).abs().max().item() <= 2e-5
        assert (feats[i, L:] == 777.0).all(), i
    # float64 pool (frames as the extractor returns them), identity rows
    p64 = pool[:n].to(torch.float64).contiguous()
    f2 = torch.zeros_like(feats)
    k.forward_rows_torch(p64, None, counts, f2, fmax)
    for i in (0, 2, n - 1):
        L = int(counts[i])
        assert torch.equal(f2[i, :L], one(pool[i, :L][None])[0])
    with pytest.raises(Exception):
        k.forward_rows_torch(pool, rows, counts + fmax, feats, fmax)          # counts beyond the call's frames
