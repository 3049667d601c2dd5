"""CPU checks of the LPCNet oracle: self-generated golden vectors (regression pin; parity with xiph is
UNPINNED, see oracle/lpcnet_oracle.c), known-answer pieces of the published algorithm, and structure."""
import hashlib

import numpy as np

from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features


def test_golden_regression(oracle, golden):
    g = golden("lpcnet_self.npz")
    blob = synthetic_blob(0)
    assert np.array_equal(np.frombuffer(hashlib.sha256(blob).digest(), dtype=np.uint8), g["blob_sha"])
    m = oracle.lpcnet_model(blob)
    f = synthetic_features(2, 30)
    assert np.array_equal(oracle.lpcnet_utterance(m, f), g["utt2_pcm"])


def test_first_two_frames_are_silent_and_state_persists(oracle):
    m = oracle.lpcnet_model(synthetic_blob(0))
    f = synthetic_features(5, 8)
    whole = oracle.lpcnet_utterance(m, f)
    assert not whole[:320].any() and whole[320:].any()        # FEATURES_DELAY = 2 frames of look-ahead
    dec = oracle.decoder(m)
    parts = np.concatenate([dec.synthesize(f[i]) for i in range(8)])
    assert np.array_equal(parts, whole)
    dec.reset()
    assert np.array_equal(np.concatenate([dec.synthesize(f[i]) for i in range(3)]), whole[:480])


def test_known_answers_of_the_published_tables(oracle):
    m = oracle.lpcnet_model(synthetic_blob(0))
    tansig = oracle.lpcnet_table(m, 0, 201)
    assert tansig[0] == 0 and abs(tansig[25] - np.tanh(1.0)) < 1e-6 and tansig[200] == 1.0
    logit = oracle.lpcnet_table(m, 1, 256)
    p = 1 / (1 + np.exp(-logit.astype(np.float64)))
    assert abs(p[0] - 0.025) < 1e-6 and abs(p[255] - 0.975) < 1e-6       # lpcnet_init(): .025 + .95*i/255
    u2l = oracle.lpcnet_table(m, 2, 256)
    assert u2l[128] == 0 and abs(u2l[255] - 32768 * (256 ** (127 / 128.) - 1) / 255) < 0.5 and np.all(np.diff(u2l) > 0)


def test_lpc_from_cepstrum_is_a_stable_predictor(oracle):
    m = oracle.lpcnet_model(synthetic_blob(0))
    for seed in range(5):
        cep = synthetic_features(seed, 1)[0, :18]
        lpc = oracle.lpc_from_cepstrum(m, cep)
        roots = np.roots(np.concatenate([[1.0], lpc.astype(np.float64)]))
        assert np.all(np.abs(roots) < 1.0)                                   # Levinson on a valid autocorrelation
