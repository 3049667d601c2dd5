"""GPU parity of the LPCNet path, through the C ABI, against the CPU oracle (same seeded inputs) and the
committed golden vectors.  Integer outputs (mu-law excitation index, int16 PCM) must be bit-exact; float
taps (conditioning vectors, LPC, pre-quantised sample value) are compared with array_equal too, i.e. at
tolerance 0 -- north_star allows +-1 LSB PCM / a float tolerance on the excitation; this build meets the
stricter bar because it keeps the reference's operation order."""
import numpy as np
import pytest

from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(oracle):
    from dss_amd import lpcnet
    blob = synthetic_blob(0)
    lpcnet.load_model(blob)
    return oracle.lpcnet_model(blob)


@pytest.mark.parametrize("B,F", [(3, 12), (1, 1), (13, 100), (25, 101)])
def test_frame_network_taps_bit_exact(oracle, model, B, F):
    """The frame-rate network's three outputs per frame against the oracle, at row counts that take each of the dense layers'
    block shapes (2 rows per block up to 1024 rows, 4 up to 2048, 8 beyond) and an odd number of frames (frame_lpc_kernel
    takes two per block)."""
    from dss_amd.lpcnet import LPCNetBatch
    feats = np.stack([synthetic_features(40 + b, F) for b in range(B)])
    gpu = LPCNetBatch(B, F)
    gpu.synthesize(feats)
    for b in sorted({0, B // 2, B - 1}):
        dec = oracle.decoder(model)
        for t in range(F):
            dec.frame_network(feats[b, t])
            assert np.array_equal(gpu.tap(b, 2, F)[t], dec.tap(2, 16)), (b, t, "lpc")
            assert np.array_equal(gpu.tap(b, 0, F)[t], dec.tap(0, 1152)), (b, t, "gru_a_condition")
            assert np.array_equal(gpu.tap(b, 1, F)[t], dec.tap(1, 48)), (b, t, "gru_b_condition")


def test_free_running_excitation_and_pcm_bit_exact(oracle, model):
    from dss_amd.lpcnet import LPCNetBatch
    B, F = 4, 20
    feats = np.stack([synthetic_features(60 + b, F) for b in range(B)])
    gpu = LPCNetBatch(B, F)
    gpu.enable_trace(True)
    pcm = gpu.synthesize(feats)
    for b in range(B):
        dec = oracle.decoder(model, trace_cap=F * 160)
        want = np.concatenate([dec.synthesize(feats[b, t]) for t in range(F)])
        n = (F - 2) * 160
        exc = gpu.tap(b, 3, F).reshape(-1)[320:]
        pre = gpu.tap(b, 4, F).reshape(-1)[320:]
        assert np.array_equal(exc.astype(np.uint8), dec.trace_exc[:n]), b       # mu-law index: exact
        assert np.array_equal(pre, dec.trace_pcm[:n]), b                         # pre-quantised value: tol 0
        assert np.array_equal(pcm[b], want), b                                   # int16 PCM: exact (<= +-1 LSB bar)


def test_golden_utterances(golden, model):
    from dss_amd.lpcnet import LPCNetBatch
    g = golden("lpcnet_self.npz")
    feats = np.stack([synthetic_features(0, 100), synthetic_features(1, 100)])
    pcm = LPCNetBatch(2, 100).synthesize(feats)
    assert np.array_equal(pcm[0], g["utt0_pcm"]) and np.array_equal(pcm[1], g["utt1_pcm"])
    short = LPCNetBatch(1, 30).synthesize(synthetic_features(2, 30)[None])
    assert np.array_equal(short[0], g["utt2_pcm"])


def test_state_persists_across_calls_and_reset(golden, model):
    from dss_amd.lpcnet import LPCNetBatch
    g = golden("lpcnet_self.npz")
    f = synthetic_features(2, 30)
    gpu = LPCNetBatch(2, 7)
    chunks = []
    for a in range(0, 30, 7):                                    # ragged chunking: 7,7,7,7,2 frames
        part = np.stack([f[a:a + 7], f[a:a + 7]])
        chunks.append(gpu.synthesize(part))
    got = np.concatenate(chunks, axis=1)
    assert np.array_equal(got[0], g["utt2_pcm"]) and np.array_equal(got[1], g["utt2_pcm"])
    gpu.reset(1)                                                 # only slot 1 restarts
    again = gpu.synthesize(np.stack([f[:7], f[:7]]))
    assert np.array_equal(again[1], g["utt2_pcm"][:7 * 160])
    assert not np.array_equal(again[0], g["utt2_pcm"][:7 * 160])


def test_feature_stride_36_and_shape_errors(golden, model):
    from dss_amd import _lib
    from dss_amd.lpcnet import LPCNetBatch
    g = golden("lpcnet_self.npz")
    f36 = np.zeros((1, 30, 36), np.float32)
    f36[0, :, :20] = synthetic_features(2, 30)
    f36[0, :, 20:] = 123.0                                       # the 16 trailing floats are ignored (LPCNet.pyx:115)
    gpu = LPCNetBatch(1, 30)
    assert np.array_equal(gpu.synthesize(f36)[0], g["utt2_pcm"])
    with pytest.raises(_lib.DssError):
        gpu.synthesize(np.zeros((2, 30, 20), np.float32))        # more utterances than slots
    with pytest.raises(ValueError):
        gpu.synthesize(np.zeros((1, 30, 19), np.float32))


def test_dropin_lpcnet_class(oracle, model):
    import LPCNet
    f = synthetic_features(9, 6)
    net = LPCNet.LPCNet()
    assert net.LPCNET_FRAME_SIZE == 160
    dec = oracle.decoder(model)
    for t in range(6):
        out = net.synthesize(f[t, :])
        assert out.dtype == np.int16 and out.shape == (160,)
        assert np.array_equal(out, dec.synthesize(f[t]))
    net.reset_decoder()
    dec.reset()
    assert np.array_equal(net.synthesize(f[0]), dec.synthesize(f[0]))
    with pytest.raises(ValueError):
        net.synthesize(f[0].astype(np.float64))
    with pytest.raises(ValueError):
        net.synthesize(f[:2])
    # DelayedLPCNetVocoder.synthesize's loop (local/units.py:531-538) over a segment
    net2 = LPCNet.LPCNet()
    seg = np.hstack([net2.synthesize(row) for row in f.astype(np.float32)])
    assert seg.shape == (960,)


def test_full_size_batch_256_properties(golden, model):
    """BASELINE config 2 size: 256 x 1-s utterances.  Size-independent properties: utterances are
    independent of their slot and of their neighbours, runs are deterministic, known utterances match golden."""
    import torch
    from dss_amd.lpcnet import LPCNetBatch
    g = golden("lpcnet_self.npz")
    B, F = 256, 100
    feats = np.stack([synthetic_features(b, F) for b in range(B)])
    gpu = LPCNetBatch(B, F)
    pcm = gpu.synthesize(feats)
    assert np.array_equal(pcm[0], g["utt0_pcm"]) and np.array_equal(pcm[1], g["utt1_pcm"])
    perm = np.random.default_rng(1).permutation(B)
    gpu.reset()
    d = torch.from_numpy(feats[perm]).cuda()
    pcm2 = gpu.synthesize_torch(d).cpu().numpy()                 # device-resident entry point, permuted slots
    assert np.array_equal(pcm2, pcm[perm])
    assert (pcm[:, :320] == 0).all() and np.abs(pcm[:, 320:]).max() > 1000


def _oracle_pcm(oracle, blob, feats):
    m = oracle.lpcnet_model(blob)
    return np.stack([oracle.lpcnet_utterance(m, f) for f in feats])


def test_other_sparsity_patterns_and_generic_fallback(oracle):
    """Models with different block-sparsity patterns: seed 1 has row groups with 11-12 z/r blocks (the register
    slot capacity), a denser model exceeds the CU-resident kernel's capacities and must run on the generic
    kernel -- all bit-exact.  Also forces the generic kernel on the default model."""
    from dss_amd import lpcnet
    from dss_amd.lpcnet import LPCNetBatch
    from dss_amd.lpcnet_weights import make_synthetic_weights, pack_blob
    feats = np.stack([synthetic_features(700 + b, 6) for b in range(3)])
    try:
        for seed, density in ((1, (0.05, 0.05, 0.20)), (2, (0.05, 0.05, 0.20)), (3, (0.12, 0.10, 0.30))):
            blob = pack_blob(make_synthetic_weights(seed, density=density))
            lpcnet.load_model(blob)
            got = LPCNetBatch(3, 6).synthesize(feats)
            assert np.array_equal(got, _oracle_pcm(oracle, blob, feats)), (seed, density)
        blob = synthetic_blob(0)
        lpcnet.load_model(blob)
        gen = LPCNetBatch(3, 6)
        gen.enable_trace(16)                                   # development switch: force the generic kernel
        assert np.array_equal(gen.synthesize(feats), _oracle_pcm(oracle, blob, feats))
    finally:
        lpcnet.load_model(synthetic_blob(0))


def test_batch_larger_than_the_chip(golden, model):
    """More utterances than CUs (config 4's per-GPU share is 1024): workgroups run in several rounds."""
    from dss_amd.lpcnet import LPCNetBatch
    g = golden("lpcnet_self.npz")
    B, F = 600, 30
    feats = np.stack([synthetic_features(2, F)] * B)
    feats[1::2] = synthetic_features(5, F)
    pcm = LPCNetBatch(B, F).synthesize(feats)
    assert np.array_equal(pcm[0], g["utt2_pcm"]) and np.array_equal(pcm[598], g["utt2_pcm"])
    assert (pcm[1::2] == pcm[1]).all() and not np.array_equal(pcm[0], pcm[1])


def test_ragged_rows_and_slot_indexed_state(oracle, model):
    """dss_lpcnet_batch_synthesize_ragged: rows of different lengths in one launch, each continuing the decoder slot it
    names -- the shape of the reference's real callers (files of different lengths, local/training.py:182-198;
    segments finishing on some streams while the vocoder state carries across segments, local/units.py:524,531-538).
    Checked per slot against one oracle decoder fed the same frames in the same order; a zero-frame row leaves its
    slot untouched; both the fast and the generic kernel."""
    from dss_amd.lpcnet import LPCNetBatch
    calls = [                                         # (slot, first frame, number of frames) per row
        [(3, 0, 2), (0, 0, 5), (4, 0, 1)],
        [(0, 5, 4), (3, 2, 0), (1, 0, 3), (4, 1, 6)],
        [(3, 2, 7), (1, 3, 1)],
    ]
    feats = {s: synthetic_features(900 + s, 12) for s in range(5)}
    for force_generic in (False, True):
        gpu = LPCNetBatch(5, 7)
        if force_generic:
            gpu.enable_trace(16)
        decs = {s: oracle.decoder(model) for s in range(5)}
        for rows in calls:
            got = gpu.synthesize_ragged([feats[s][a:a + n] for s, a, n in rows], slots=[s for s, _, _ in rows])
            for (s, a, n), pcm in zip(rows, got):
                want = [decs[s].synthesize(feats[s][t]) for t in range(a, a + n)]
                want = np.concatenate(want) if want else np.empty(0, np.int16)
                assert pcm.shape == (n * 160,) and np.array_equal(pcm, want), (force_generic, s, a, n)


def test_ragged_rows_on_the_pair_kernel(oracle, model):
    """Ragged calls on the two-utterances-per-workgroup kernel (RAGGED instantiation of lpcnet_sample_pair_kernel; chosen
    automatically beyond one row per CU, forced here): rows 2k and 2k+1 share a workgroup, run packed over the frames both
    have, and the longer one finishes alone from the state the packed part left in global memory.  Same call shapes as
    the reference's bulk callers (local/training.py:182-198, local/units.py:524,531-538).  Rows in the caller's order
    (no sorting), so that the pairs are the ones written here: longer first, longer second, equal, a zero-frame partner,
    a lone last row; later calls continue slots with frame_count > 0 next to fresh ones (different numbers of silent
    frames: such a pair runs one row after the other).  One oracle decoder per slot is the reference."""
    from dss_amd.lpcnet import LPCNetBatch
    calls = [                                         # (slot, first frame, number of frames) per row
        [(3, 0, 6), (0, 0, 2), (4, 0, 1), (1, 0, 5), (2, 0, 3), (5, 0, 3), (6, 0, 0), (7, 0, 4), (8, 0, 2)],
        [(0, 2, 4), (9, 0, 4), (3, 6, 2), (1, 5, 6), (6, 0, 3), (4, 1, 0), (2, 3, 1)],
        [(9, 4, 7), (8, 2, 7), (7, 4, 1)],
    ]
    feats = {s: synthetic_features(950 + s, 12) for s in range(10)}
    for trace in (False, True):
        gpu = LPCNetBatch(10, 7)
        gpu.set_multi(2)
        if trace:
            gpu.enable_trace(True)
        decs = {s: oracle.decoder(model) for s in range(10)}
        for rows in calls:
            got = gpu.synthesize_ragged([feats[s][a:a + n] for s, a, n in rows], slots=[s for s, _, _ in rows], longest_first=False)
            for (s, a, n), pcm in zip(rows, got):
                want = [decs[s].synthesize(feats[s][t]) for t in range(a, a + n)]
                want = np.concatenate(want) if want else np.empty(0, np.int16)
                assert pcm.shape == (n * 160,) and np.array_equal(pcm, want), (trace, s, a, n)
    # the offline shape at a size where the rule picks the pair kernel by itself: 600 files, longest first
    n = 600
    lengths = [12 if i % 7 == 0 else 2 + (i % 9) for i in range(n)]
    f2 = synthetic_features(2, 12)
    want = oracle.lpcnet_utterance(model, f2)
    got = LPCNetBatch(n, 12).synthesize_ragged([f2[:k] for k in lengths])
    for k, pcm in zip(lengths, got):
        assert np.array_equal(pcm, want[:k * 160])


def test_ragged_files_fresh_state_longest_first(golden, model):
    """Offline shape: more files than CUs, different lengths, fresh decoder each; rows are dispatched longest first
    and returned in the caller's order."""
    from dss_amd.lpcnet import LPCNetBatch
    g = golden("lpcnet_self.npz")
    n = 300
    lengths = [30 if i % 7 == 0 else 3 + (i % 11) for i in range(n)]
    f2 = synthetic_features(2, 30)
    got = LPCNetBatch(n, 30).synthesize_ragged([f2[:k] for k in lengths])
    for k, pcm in zip(lengths, got):
        assert np.array_equal(pcm, g["utt2_pcm"][:k * 160])


def test_ragged_argument_errors(model):
    from dss_amd import _lib
    from dss_amd.lpcnet import LPCNetBatch
    gpu = LPCNetBatch(4, 5)
    f = synthetic_features(1, 5)
    with pytest.raises(_lib.DssError, match="twice"):
        gpu.synthesize_ragged([f, f], slots=[2, 2])
    with pytest.raises(_lib.DssError, match="out of range"):
        gpu.synthesize_ragged([f], slots=[4])
    with pytest.raises(_lib.DssError, match="frames"):
        gpu.synthesize_ragged([synthetic_features(1, 6)])


def test_long_utterance_and_extreme_features(oracle, model):
    """A 3-second utterance (no drift over 48 000 samples) and features at the edges of what the frame network accepts:
    pitch values that clamp to index 33 and 255, large cepstra (LPC recursion under stress, tanh/sigmoid tables at their
    clamps) and an all-zero row -- all bit-exact against the oracle."""
    from dss_amd.lpcnet import LPCNetBatch
    F = 300
    long_f = synthetic_features(4242, F)
    edge = synthetic_features(4243, 40)
    edge[5:10, 18] = -10.0                          # pitch index clamps to 33
    edge[10:15, 18] = 10.0                          # ... and to 255
    edge[15:20, :18] *= 6.0                         # hot cepstrum
    edge[20:25, :] = 0.0
    edge[25:30, 19] = 3.0
    got_long = LPCNetBatch(1, F).synthesize(long_f[None])[0]
    dec = oracle.decoder(model)
    want_long = np.concatenate([dec.synthesize(long_f[t]) for t in range(F)])
    assert np.array_equal(got_long, want_long)
    got_edge = LPCNetBatch(1, 40).synthesize(edge[None])[0]
    dec = oracle.decoder(model)
    want_edge = np.concatenate([dec.synthesize(edge[t]) for t in range(40)])
    assert np.array_equal(got_edge, want_edge)


def test_config4_per_gpu_share_1024_utterances(golden, model):
    """BASELINE.json configs[3] per GPU: 1024 utterances of 1 s in one call (four rounds of workgroups).  Size-independent
    properties: identical inputs give identical PCM wherever they sit in the batch, and the two golden utterances come
    out bit-exact from the first and the last round."""
    from dss_amd.lpcnet import LPCNetBatch
    g = golden("lpcnet_self.npz")
    B, F = 1024, 100
    f0, f1 = synthetic_features(0, F), synthetic_features(1, F)
    feats = np.empty((B, F, 20), dtype=np.float32)
    feats[0::2] = f0
    feats[1::2] = f1
    pcm = LPCNetBatch(B, F).synthesize(feats)
    assert np.array_equal(pcm[0], g["utt0_pcm"]) and np.array_equal(pcm[1], g["utt1_pcm"])
    assert np.array_equal(pcm[1022], g["utt0_pcm"]) and np.array_equal(pcm[1023], g["utt1_pcm"])
    assert (pcm[0::2] == pcm[0]).all() and (pcm[1::2] == pcm[1]).all()


# ---------------------------------------------------------------------------------------------------------------
# round 2: teacher-forced twin of tests/test_oracle_lpcnet_pins.py, the GRU A association flag, the device pow
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("generic", [False, True])
def test_teacher_forced_logits_match_the_oracle(oracle, model, generic):
    """SURVEY 7 hard-part 2, teacher-forced mode: both sides are fed the SAME excitation sequence, so one flipped
    comparison cannot hide everything after it; all 255 node logits of every sample are compared.  Stated tolerance:
    0 (array_equal) -- the kernels keep the oracle's operation order; the oracle's own logits are pinned against the
    layer definitions within 2e-3 on the CPU (test_sample_step_teacher_forced_against_layer_definitions)."""
    from dss_amd.lpcnet import LPCNetBatch
    B, F = 2, 6
    n = F * 160
    feats = np.stack([synthetic_features(300 + b, F) for b in range(B)])
    rng = np.random.default_rng(17)
    exc = np.clip(np.rint(128 + rng.normal(0, 30, (B, n))), 0, 255).astype(np.uint8)
    gpu = LPCNetBatch(B, F)
    gpu.enable_trace(17 if generic else 1)
    gpu.force_excitation(exc, F)
    pcm = gpu.synthesize(feats)
    for b in range(B):
        dec = oracle.decoder(model, trace_cap=n)
        dec.force(exc[b, 320:])                    # the first two frames are silent: no sample step runs in them
        want = np.concatenate([dec.synthesize(feats[b, t]) for t in range(F)])
        got_exc = gpu.tap(b, 3, F).reshape(-1)[320:].astype(np.uint8)
        assert np.array_equal(got_exc, exc[b, 320:]) and np.array_equal(dec.trace_exc[:n - 320], exc[b, 320:])
        logits = gpu.tap(b, 5, F).reshape(n, 256)[320:]
        assert np.array_equal(logits, dec.forced_logits), (b, np.abs(logits - dec.forced_logits).max())
        assert np.array_equal(gpu.tap(b, 4, F).reshape(-1)[320:], dec.trace_pcm[:n - 320])
        assert np.array_equal(pcm[b], want)
    # back to free running on the same object: sampled excitations again, bit-exact
    gpu.force_excitation(None, F)
    gpu.reset()
    pcm = gpu.synthesize(feats)
    m2 = oracle.lpcnet_utterance(model, feats[0])
    assert np.array_equal(pcm[0], m2)


def test_teacher_forcing_refuses_another_shape(model):
    """the forced excitation and the logit trace are sized for the shape given to force_excitation; a call of
    any other shape (or a ragged one) while forcing is on must fail instead of indexing past them."""
    from dss_amd import _lib
    from dss_amd.lpcnet import LPCNetBatch
    gpu = LPCNetBatch(3, 6)
    gpu.enable_trace(1)
    gpu.force_excitation(np.full((2, 4 * 160), 128, np.uint8), 4)
    f = np.stack([synthetic_features(50 + b, 6) for b in range(3)])
    with pytest.raises(_lib.DssError, match="teacher forcing was set up"):
        gpu.synthesize(f)                                    # 3 x 6 against buffers for 2 x 4
    with pytest.raises(_lib.DssError, match="teacher forcing was set up"):
        gpu.synthesize_ragged([f[0, :4], f[1, :2]])
    assert gpu.synthesize(f[:2, :4]).shape == (2, 640)       # the shape it was set up for still runs
    gpu.force_excitation(None, 4)
    assert gpu.synthesize(f).shape == (3, 960)


def test_gru_a_recurrent_first_order_flag(oracle):
    """dss_blob_header.gru_a_order = 1 (xiph nnet.c 2019-20 association): both kernels follow the oracle bit for bit,
    and the teacher-forced states differ from the default order's in the last bits only."""
    from dss_amd import lpcnet
    from dss_amd.lpcnet import LPCNetBatch
    from dss_amd.lpcnet_weights import GRUA_RECUR_FIRST
    feats = np.stack([synthetic_features(800 + b, 8) for b in range(3)])
    blob = synthetic_blob(0, gru_a_order=GRUA_RECUR_FIRST)
    try:
        lpcnet.load_model(blob)
        assert lpcnet.model_info()["gru_a_order"] == 1
        want = _oracle_pcm(oracle, blob, feats)
        for generic in (False, True):
            gpu = LPCNetBatch(3, 8)
            gpu.enable_trace(17 if generic else 1)
            assert np.array_equal(gpu.synthesize(feats), want), generic
        # teacher-forced: the two orders are different float programs (logits differ somewhere) of one real function
        exc = np.full((1, 8 * 160), 131, np.uint8)
        logits = {}
        for order in (0, 1):
            b2 = synthetic_blob(0, gru_a_order=order)
            lpcnet.load_model(b2)
            gpu = LPCNetBatch(1, 8)
            gpu.enable_trace(1)
            gpu.force_excitation(exc, 8)
            gpu.synthesize(feats[:1])
            logits[order] = gpu.tap(0, 5, 8).reshape(-1, 256)[320:]
            dec = oracle.decoder(oracle.lpcnet_model(b2))
            dec.force(exc[0, 320:])
            for t in range(8):
                dec.synthesize(feats[0, t])
            assert np.array_equal(logits[order], dec.forced_logits), order
        assert not np.array_equal(logits[0], logits[1]) and np.abs(logits[0] - logits[1]).max() < 1e-3
    finally:
        lpcnet.load_model(synthetic_blob(0))


def test_skewed_sparsity_selects_a_kernel_loudly(oracle):
    """magnitude pruning gives skewed per-row-group block counts.  Moderately skewed models stay on the
    CU-resident kernel through its extended paths (z/r blocks beyond the register slots and h slots beyond 28 come from
    LDS records: model_info fast_path 2); only a model beyond the outer capacities falls to the generic kernel, with a
    RuntimeWarning.  Whatever the kernel, the choice is visible and the output is bit-exact."""
    import warnings
    from dss_amd import lpcnet
    from dss_amd.lpcnet import LPCNetBatch
    feats = np.stack([synthetic_features(820 + b, 6) for b in range(3)])
    try:
        seen = {}
        # per-group maxima (z-r / h blocks): 12/29, 16/34, 26/35, 49/51 against 12 (8) / 28 register-held slots
        for seed, skew in ((0, 0.02), (7, 0.05), (0, 0.1), (7, 0.3)):
            blob = synthetic_blob(seed, skew=skew)
            lpcnet.load_model(blob)
            info = lpcnet.model_info()
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                gpu = LPCNetBatch(3, 6)
            assert bool(w) == (not info["fast_path"])
            if w:
                assert "generic kernel" in str(w[0].message)
            want = _oracle_pcm(oracle, blob, feats)
            assert np.array_equal(gpu.synthesize(feats), want), (skew, info)
            # state carried across calls: 4 + 2 frames equal 6 in one call
            gpu.reset()
            got = np.concatenate([gpu.synthesize(feats[:, :4]), gpu.synthesize(feats[:, 4:])], axis=1)
            assert np.array_equal(got, want), (skew, info)
            seen[skew] = info["fast_path"]
        assert seen == {0.02: 2, 0.05: 2, 0.1: 2, 0.3: 0}, seen
    finally:
        lpcnet.load_model(synthetic_blob(0))


def test_extended_paths_ragged_and_teacher_forced(oracle):
    """The tail / long-list paths in the other instantiations of the CU-resident kernel: a ragged call (rows with their
    own frame counts and decoder slots) and the teacher-forced trace build, on a skewed model."""
    from dss_amd import lpcnet
    from dss_amd.lpcnet import LPCNetBatch
    blob = synthetic_blob(0, skew=0.1)
    try:
        lpcnet.load_model(blob)
        assert lpcnet.model_info()["fast_path"] == 2
        counts = [6, 3, 5]
        feats = np.stack([synthetic_features(840 + b, 6) for b in range(3)])
        gpu = LPCNetBatch(3, 6)
        got = gpu.synthesize_ragged([feats[b, :counts[b]] for b in range(3)])
        want = _oracle_pcm(oracle, blob, feats)
        for b in range(3):
            assert np.array_equal(got[b], want[b, :counts[b] * 160]), b
        # teacher forcing: every logit of every tree node under a forced excitation
        exc = np.random.default_rng(5).integers(0, 256, (1, 5 * 160), dtype=np.uint8)
        g1 = LPCNetBatch(1, 5)
        g1.enable_trace(1)
        g1.force_excitation(exc, 5)
        g1.synthesize(feats[:1, :5])
        logits = g1.tap(0, 5, 5).reshape(-1, 256)[320:]
        dec = oracle.decoder(oracle.lpcnet_model(blob))
        dec.force(exc[0, 320:])
        for t in range(5):
            dec.synthesize(feats[0, t])
        assert np.array_equal(logits, dec.forced_logits)
        # the other association order of GRU A (tail sums sit before the input term there)
        blob1 = synthetic_blob(7, gru_a_order=1, skew=0.05)
        lpcnet.load_model(blob1)
        assert lpcnet.model_info()["fast_path"] == 2
        assert np.array_equal(LPCNetBatch(3, 6).synthesize(feats), _oracle_pcm(oracle, blob1, feats))
    finally:
        lpcnet.load_model(synthetic_blob(0))


def test_device_pow_equals_host_libm_after_the_float_rounding(oracle):
    """frame_lpc_kernel evaluates (float)(pow(10., e) * compensation) on the device, everything else transcendental on
    this path is a host-built table.  Sweep the reachable exponent range densely: 1.2e7 float32 exponents in [-9, 9]
    (|cepstrum| up to ~25 after the idct scale) against glibc's pow through the C oracle's libm."""
    import ctypes
    from dss_amd import _lib
    L = _lib.require_gpu()
    n = 12_000_000
    rng = np.random.default_rng(123)
    x = np.concatenate([np.linspace(-9, 9, n // 2, dtype=np.float64), rng.uniform(-9, 9, n - n // 2)]).astype(np.float32)
    comp_vals = np.array([0.8, 1, 1, 1, 1, 1, 1, 1, 0.666667, 0.5, 0.5, 0.5, 0.333333, 0.25, 0.25, 0.2, 0.166667, 0.173913],
                         dtype=np.float32)
    comp = comp_vals[rng.integers(0, 18, n)]
    got = np.empty(n, np.float32)
    _lib.check(L.dss_selftest_exp10(x.ctypes.data, comp.ctypes.data, got.ctypes.data, n))
    want = np.empty(n, np.float32)
    oracle.lib.oracle_exp10_comp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    oracle.lib.oracle_exp10_comp(x.ctypes.data, comp.ctypes.data, want.ctypes.data, n)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad.size, x[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_device_lin2ulaw_equals_the_c_form_on_a_sweep_of_all_bit_patterns(oracle):
    """csrc/lpcnet_device.h evaluates xiph's lin2ulaw() (common.h; used by lpcnet_synthesize_tail_impl for the two mu-law
    indices of every sample, here inside the speculation) in a shorter instruction sequence: constant division as a product
    with two fused corrections, max/min clamp, float rounding.  tools/verify/lin2ulaw_exhaustive.c proves a C restatement
    of that sequence equal to the C form for every fp32 input; this closes the loop on the device itself: every 97th bit
    pattern of the whole 2^32 range (44 M values, NaNs skipped: (int) of a NaN is undefined in the C source), plus every
    pattern of the two binades around +-32768 where the PCM values live."""
    import ctypes
    from dss_amd import _lib
    L = _lib.require_gpu()
    oracle.lib.oracle_lin2ulaw_sweep.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_long, ctypes.c_void_p]
    oracle.lib.oracle_lin2ulaw_sweep.restype = None
    sweeps = [(0, 97, (1 << 32) // 97)]
    for sign in (0, 0x80000000):
        sweeps.append((sign | 0x46000000, 1, 1 << 24))            # 8192 <= |x| < 32768 (2 binades), every value
    for start, stride, n in sweeps:
        got = np.empty(n, np.uint8)
        want = np.empty(n, np.uint8)
        _lib.check(L.dss_selftest_lin2ulaw(start, stride, n, got.ctypes.data))
        oracle.lib.oracle_lin2ulaw_sweep(start, stride, n, want.ctypes.data)
        bits = (np.uint32(start) + np.arange(n, dtype=np.uint32) * np.uint32(stride))
        ok = ~np.isnan(bits.view(np.float32))
        bad = np.nonzero((got != want) & ok)[0]
        assert bad.size == 0, (start, stride, bad.size, bits[bad[:5]], got[bad[:5]], want[bad[:5]])


def test_void_synthesize_never_aborts(model):
    """cLPCNet.pxd:13 has no error channel: a bad call zero-fills the frame and is counted; the state keeps working."""
    import LPCNet
    from dss_amd import _lib
    L = _lib.load()
    net = LPCNet.LPCNet()
    f = synthetic_features(9, 4)
    before = L.dss_error_count()
    out = np.ones(80, dtype=np.int16)
    L.lpcnet_synthesize(net._st, f[0].ctypes.data, out.ctypes.data, 80)            # N != 160
    assert not out.any() and L.dss_error_count() == before + 1 and b"must be 160" in L.dss_last_error()
    assert net.synthesize(f[0]).shape == (160,)                                     # the object is still usable


# ---------------------------------------------------------------------------------------------------------------
# throughput kernel: two utterances per workgroup as the halves of packed fp32 instructions (csrc/lpcnet_sample_pair.hip)
# ---------------------------------------------------------------------------------------------------------------
def test_pair_kernel_bit_exact(oracle, model):
    """Forced on small batches: even and odd utterance counts (the last workgroup of an odd call carries one utterance),
    a single utterance, a one-frame call, and state carried over several calls (the later calls start with
    frame_count > 0, so their first frames are NOT silent)."""
    from dss_amd.lpcnet import LPCNetBatch
    for B, F in ((6, 6), (7, 5), (1, 4), (3, 1)):
        feats = np.stack([synthetic_features(1200 + b, F) for b in range(B)])
        gpu = LPCNetBatch(B, F)
        gpu.set_multi(2)
        pcm = gpu.synthesize(feats)
        for b in range(B):
            assert np.array_equal(pcm[b], oracle.lpcnet_utterance(model, feats[b])), (B, F, b)
    # a long call: 3 utterances x 2 s (32 000 sample steps, 7 hand-overs of the GRU B relay in each)
    feats = np.stack([synthetic_features(1250 + b, 200) for b in range(3)])
    gpu = LPCNetBatch(3, 200)
    gpu.set_multi(2)
    pcm = gpu.synthesize(feats)
    for b in range(3):
        assert np.array_equal(pcm[b], oracle.lpcnet_utterance(model, feats[b])), b
    # chunked: 1, 1, 1, 2, 5 frames through one batch object vs one oracle decoder per utterance
    B, F = 5, 10
    feats = np.stack([synthetic_features(1300 + b, F) for b in range(B)])
    gpu = LPCNetBatch(B, 5)
    gpu.set_multi(2)
    decs = [oracle.decoder(model) for _ in range(B)]
    t = 0
    for n in (1, 1, 1, 2, 5):
        got = gpu.synthesize(feats[:, t:t + n])
        for b in range(B):
            want = np.concatenate([decs[b].synthesize(feats[b, t + q]) for q in range(n)])
            assert np.array_equal(got[b], want), (t, n, b)
        t += n
    # the latency kernel continues a state the pair kernel left, and the other way round
    more = synthetic_features(1400, 4)
    for mode, rows in ((1, slice(0, 2)), (2, slice(2, 4))):
        gpu.set_multi(mode)
        got = gpu.synthesize(np.stack([more[rows]] * B))
        for b in range(B):
            want = np.concatenate([decs[b].synthesize(more[q]) for q in range(rows.start, rows.stop)])
            assert np.array_equal(got[b], want), ("hand-over", mode, b)


def test_pair_kernel_unequal_silence_and_trace(oracle, model):
    """Two utterances of one workgroup whose decoders are at different points of the two-frame look-ahead (one fresh, one
    continued) cannot share the silent-frame schedule: the workgroup runs them one after the other.  Then the traced
    instantiation: sampled excitation, pre-quantised value, and teacher-forced logits of all 255 nodes for both
    utterances of a workgroup."""
    from dss_amd.lpcnet import LPCNetBatch
    B, F = 4, 5
    feats = np.stack([synthetic_features(1600 + b, 2 * F) for b in range(B)])
    gpu = LPCNetBatch(B, F)
    gpu.set_multi(2)
    decs = [oracle.decoder(model) for _ in range(B)]
    first = gpu.synthesize(feats[:, :F])
    for b in range(B):
        assert np.array_equal(first[b], np.concatenate([decs[b].synthesize(feats[b, t]) for t in range(F)])), b
    gpu.reset(1)                                         # utterances 0 and 1 share a workgroup; 1 starts over, 0 goes on
    gpu.reset(2)
    decs[1].reset()
    decs[2].reset()
    second = gpu.synthesize(feats[:, F:])
    for b in range(B):
        want = np.concatenate([decs[b].synthesize(feats[b, t]) for t in range(F, 2 * F)])
        assert np.array_equal(second[b], want), ("unequal silence", b)
    # trace build, free running
    B, F = 3, 6
    n = F * 160
    feats = np.stack([synthetic_features(1700 + b, F) for b in range(B)])
    gpu = LPCNetBatch(B, F)
    gpu.set_multi(2)
    gpu.enable_trace(True)
    pcm = gpu.synthesize(feats)
    for b in range(B):
        dec = oracle.decoder(model, trace_cap=n)
        want = np.concatenate([dec.synthesize(feats[b, t]) for t in range(F)])
        assert np.array_equal(gpu.tap(b, 3, F).reshape(-1)[320:].astype(np.uint8), dec.trace_exc[:n - 320]), b
        assert np.array_equal(gpu.tap(b, 4, F).reshape(-1)[320:], dec.trace_pcm[:n - 320]), b
        assert np.array_equal(pcm[b], want), b
    # teacher forced
    rng = np.random.default_rng(23)
    exc = np.clip(np.rint(128 + rng.normal(0, 30, (B, n))), 0, 255).astype(np.uint8)
    gpu.reset()
    gpu.force_excitation(exc, F)
    gpu.synthesize(feats)
    for b in range(B):
        dec = oracle.decoder(model, trace_cap=n)
        dec.force(exc[b, 320:])
        for t in range(F):
            dec.synthesize(feats[b, t])
        logits = gpu.tap(b, 5, F).reshape(n, 256)[320:]
        assert np.array_equal(logits, dec.forced_logits), (b, np.abs(logits - dec.forced_logits).max())


def test_pair_kernel_other_models_and_selection(oracle):
    """Recurrent-first association order and a model at the 12-slot z/r capacity through the pair kernel; a model that
    needs the extended paths refuses the forced choice (it stays on the latency kernel)."""
    from dss_amd import _lib, lpcnet
    from dss_amd.lpcnet import LPCNetBatch
    from dss_amd.lpcnet_weights import make_synthetic_weights, pack_blob
    feats = np.stack([synthetic_features(1500 + b, 5) for b in range(9)])
    try:
        for blob in (synthetic_blob(0, gru_a_order=1), pack_blob(make_synthetic_weights(1))):
            lpcnet.load_model(blob)
            gpu = LPCNetBatch(9, 5)
            gpu.set_multi(2)
            want = _oracle_pcm(oracle, blob, feats)
            assert np.array_equal(gpu.synthesize(feats), want)
            # the same rows as a ragged call (RAGGED instantiation of this model's kernel): fresh decoders, lengths 5 .. 1
            gpu.reset()
            lens = [5, 2, 4, 4, 1, 3, 5, 0, 2]
            got = gpu.synthesize_ragged([feats[b, :n] for b, n in enumerate(lens)], longest_first=False)
            for b, n in enumerate(lens):
                assert np.array_equal(got[b], want[b, :n * 160]), (b, n)
        lpcnet.load_model(synthetic_blob(0, skew=0.1))
        gpu = LPCNetBatch(9, 5)
        with pytest.raises(_lib.DssError, match="do not fit"):
            gpu.set_multi(2)
    finally:
        lpcnet.load_model(synthetic_blob(0))


def test_pair_kernel_is_chosen_beyond_one_utterance_per_cu(golden, model):
    """601 utterances on 256 CUs: the automatic choice (one full round of 512 rows on the pair kernel, the remaining 89 as
    one round of the one-utterance kernel, which starts at row 512: b.utt0), the forced pair kernel (an odd count: its last
    workgroup is half filled) and the forced latency kernel agree bit for bit with each other and with the golden utterance."""
    from dss_amd.lpcnet import LPCNetBatch
    g = golden("lpcnet_self.npz")
    B, F = 601, 30
    feats = np.stack([synthetic_features(2, F)] * B)
    feats[1::2] = synthetic_features(5, F)
    gpu = LPCNetBatch(B, F)
    gpu.enable_timing(True)
    auto = gpu.synthesize(feats)
    t_auto = gpu.kernel_ms(0)
    gpu.reset()
    gpu.set_multi(1)
    gpu.enable_timing(True)
    one = gpu.synthesize(feats)
    t_one = gpu.kernel_ms(0)
    gpu.reset()
    gpu.set_multi(2)
    gpu.enable_timing(True)
    two = gpu.synthesize(feats)
    t_two = gpu.kernel_ms(0)
    assert np.array_equal(auto, one) and np.array_equal(auto, two)
    assert np.array_equal(auto[0], g["utt2_pcm"]) and np.array_equal(auto[600], g["utt2_pcm"])
    print(f"601 x {F} frames: auto {t_auto:.1f} ms, one utterance per workgroup {t_one:.1f} ms, two {t_two:.1f} ms")


def test_lanes_share_the_decoder_slots_of_their_batch(oracle, model):
    """dss_lpcnet_batch_create_lane: launch contexts with their own scratch on the decoder slots of one batch.  Two lanes on two
    streams synthesise DIFFERENT slots at the same time (nothing between the launches orders them), then continue each
    other's slots after an event: per slot the PCM is what one oracle decoder gives for the same frames in the same order.
    More rows than the lane has, an unknown slot, a lane of a lane are refused; a parent destroyed first lives on in its lanes."""
    import torch
    from dss_amd import _lib
    from dss_amd.lpcnet import LPCNetBatch
    L = _lib.require_gpu()
    parent = LPCNetBatch(6, 1)
    lanes = [parent.create_lane(3, 9), parent.create_lane(3, 9)]
    streams = [L.dss_stream_create(), L.dss_stream_create()]
    ev = [L.dss_event_create(), L.dss_event_create()]
    assert all(streams) and all(ev)
    feats = {s: synthetic_features(700 + s, 20) for s in range(6)}
    decs = {s: oracle.decoder(model) for s in range(6)}
    pos = {s: 0 for s in range(6)}
    rounds = [([(0, 4), (2, 9), (4, 1)], [(1, 7), (3, 2), (5, 9)]),          # lane 0 | lane 1: (slot, frames)
              ([(1, 3), (5, 5)], [(0, 6), (4, 9), (2, 2)]),                  # the lanes swap slots
              ([(3, 9)], [(1, 1)])]
    for rnd in rounds:
        outs = []
        for li, rows in enumerate(rnd):
            n, fmax = len(rows), max(c for _, c in rows)
            f = np.zeros((n, fmax, 20), np.float32)
            for k, (s, c) in enumerate(rows):
                f[k, :c] = feats[s][pos[s]:pos[s] + c]
            ft = torch.from_numpy(f).cuda()
            torch.cuda.synchronize()
            _lib.check(L.dss_stream_wait_event(streams[li], ev[1 - li]))        # the other lane's previous round touched my slots
            pcm = lanes[li].synthesize_ragged_torch(ft, [c for _, c in rows], slots=[s for s, _ in rows], stream=streams[li])
            _lib.check(L.dss_event_record(ev[li], streams[li]))
            outs.append((rows, pcm, ft))
        for li in range(2):
            _lib.check(L.dss_stream_synchronize(streams[li]))
        for rows, pcm, _ in outs:
            got = pcm.cpu().numpy()
            for k, (s, c) in enumerate(rows):
                want = np.concatenate([decs[s].synthesize(feats[s][t]) for t in range(pos[s], pos[s] + c)])
                assert np.array_equal(got[k, :c * 160], want), (s, c)
                pos[s] += c
    f = torch.zeros((4, 2, 20), dtype=torch.float32, device="cuda")
    with pytest.raises(_lib.DssError, match="shape out of range"):
        lanes[0].synthesize_ragged_torch(f, [1] * 4, slots=[0, 1, 2, 3])
    with pytest.raises(_lib.DssError, match="slot 6 out of range"):
        lanes[0].synthesize_ragged_torch(f[:1], [1], slots=[6])
    with pytest.raises(_lib.DssError, match="not itself a lane"):
        lanes[0].create_lane(1, 1)
    # the parent's own handle goes first: the lanes keep its slots alive
    L.dss_lpcnet_batch_destroy(parent._h)
    parent._h = None
    c = 2
    ft = torch.from_numpy(feats[0][pos[0]:pos[0] + c][None].copy()).cuda()
    got = lanes[1].synthesize_ragged_torch(ft, [c], slots=[0]).cpu().numpy()
    want = np.concatenate([decs[0].synthesize(feats[0][t]) for t in range(pos[0], pos[0] + c)])
    assert np.array_equal(got[0], want)
    for ln in lanes:
        ln.close()
    for s_ in streams:
        L.dss_stream_destroy(s_)
    for e_ in ev:
        L.dss_event_destroy(e_)
