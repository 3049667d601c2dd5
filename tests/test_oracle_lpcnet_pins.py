"""Independent pins of oracle/lpcnet_oracle.c (CPU only).

The reference ships no LPCNet source, weights or golden waveform (extensions/lpcnet/LPCNet/ is an empty submodule,
.gitmodules:1-3; extensions/lpcnet/setup.py:24,34-36), so the oracle cannot be checked against xiph's own output here.
What CAN be checked is every published sub-algorithm against something that is NOT the oracle: published check
values, closed forms, numpy / scipy, and restatements written in this file from the LAYER DEFINITIONS (Keras GRU with
reset_after, Conv1D, Dense, MDense) rather than from the C text.  DESIGN.md section 2 maps each assumption of the
oracle to the test below that covers it.  The binding whose arithmetic this pins: extensions/lpcnet/cLPCNet.pxd:13.
"""
import os

import numpy as np
import pytest

from dss_amd.lpcnet_weights import (GRUA_INPUT_FIRST, GRUA_RECUR_FIRST, synthetic_blob, synthetic_features, unpack_blob)

M32 = 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------------
# (a) kiss99: Marsaglia's posting of 20 Jan 1999 ("Random numbers for C: End, at last?"), written here from the macros
#     znew/wnew/MWC/SHR3/CONG/KISS of that posting, with its own self-test values
# ---------------------------------------------------------------------------------------------------------
class Kiss99Py:
    def __init__(self, z, w, jsr, jcong):
        self.z, self.w, self.jsr, self.jcong = z, w, jsr, jcong

    def mwc(self):
        self.z = (36969 * (self.z & 65535) + (self.z >> 16)) & M32
        self.w = (18000 * (self.w & 65535) + (self.w >> 16)) & M32
        return ((self.z << 16) + self.w) & M32

    def shr3(self):
        j = self.jsr
        j ^= (j << 17) & M32
        j ^= j >> 13
        j ^= (j << 5) & M32
        self.jsr = j
        return j

    def cong(self):
        self.jcong = (69069 * self.jcong + 1234567) & M32
        return self.jcong

    def kiss(self):
        return ((self.mwc() ^ self.cong()) + self.shr3()) & M32

    def state(self):
        return np.array([self.z, self.w, self.jsr, self.jcong], dtype=np.uint32)


def test_kiss99_reproduces_marsaglias_published_check_values(oracle):
    # settable(12345,65435,34221,12345,9983651,95746118) fills t[256] with KISS (256 draws), the LFIB4 and SWB loops
    # that follow do not touch the KISS state; then "for(i=1;i<1000001;i++){k=KISS;} printf("%u", k-1372460312U)" = 0
    py = Kiss99Py(12345, 65435, 34221, 12345)
    for _ in range(256):
        py.kiss()
    k = 0
    for _ in range(1000000):
        k = py.kiss()
    assert k == 1372460312                       # this file's restatement of the posting is right ...
    ctx = oracle.kiss99(seed4=(12345, 65435, 34221, 12345))
    oracle.kiss99_draw(ctx, 256)
    assert oracle.kiss99_draw(ctx, 1000000, keep=1)[0] == 1372460312     # ... and so is the oracle's kiss99_rand
    assert np.array_equal(ctx, py.state())
    # the posting's next three lines run the components alone from where KISS left them
    for _ in range(1000000):
        k = py.cong()
    assert k == 1529210297
    for _ in range(1000000):
        k = py.shr3()
    assert k == 2642725982
    for _ in range(1000000):
        k = py.mwc()
    assert k == 904977562


def test_kiss99_lpcnet_seeding_by_hand(oracle):
    # kiss99_srand(ctx, "LPCNet", 6) of xiph's kiss99.c, stepped by hand: defaults of the posting, the first four
    # bytes are XORed into z, w, jsr, jcong and one value is drawn, the remaining two bytes go into z and w
    py = Kiss99Py(362436069, 521288629, 123456789, 380116160)
    d = b"LPCNet"
    py.z ^= d[0]; py.w ^= d[1]; py.jsr ^= d[2]; py.jcong ^= d[3]
    py.kiss()
    py.z ^= d[4]; py.w ^= d[5]
    assert py.z not in (0, 0x9068FFFF) and py.w not in (0, 0x464FFFFF) and py.jsr != 0     # short-cycle guards idle
    ctx = oracle.kiss99(srand=d)
    assert np.array_equal(ctx, py.state())
    want = [py.kiss() for _ in range(5)]
    assert oracle.kiss99_draw(ctx, 5, keep=5).tolist() == want
    # byte counts that are not 6: the tail rules (n mod 4 = 0, 1, 2, 3)
    for data in (b"LPCN", b"LPCNe", b"LPCNetX", b"abc"):
        p = Kiss99Py(362436069, 521288629, 123456789, 380116160)
        i = 3
        while i < len(data):
            p.z ^= data[i - 3]; p.w ^= data[i - 2]; p.jsr ^= data[i - 1]; p.jcong ^= data[i]
            p.kiss()
            i += 4
        if i - 3 < len(data):
            p.z ^= data[i - 3]
        if i - 2 < len(data):
            p.w ^= data[i - 2]
        if i - 1 < len(data):
            p.jsr ^= data[i - 1]
        assert np.array_equal(oracle.kiss99(srand=data), p.state()), data


# ---------------------------------------------------------------------------------------------------------
# (b) mu-law companding (xiph common.h lin2ulaw / ulaw2lin)
# ---------------------------------------------------------------------------------------------------------
def test_mulaw_round_trip_monotonicity_and_textbook_formula(oracle):
    L = oracle.lib
    codes = np.arange(256)
    lin = np.array([L.oracle_ulaw2lin(float(u)) for u in codes], dtype=np.float64)
    assert [L.oracle_lin2ulaw(float(v)) for v in lin] == codes.tolist()           # lin2ulaw(ulaw2lin(u)) == u, all u
    assert np.all(np.diff(lin) > 0) and lin[128] == 0.0
    assert np.allclose(lin[129:], -lin[127:0:-1], rtol=0, atol=0)                    # odd symmetry about code 128
    # textbook mu = 255 expansion: x = sign * (32768/255) * (256^(|u-128|/128) - 1)
    u = (codes - 128).astype(np.float64)
    want = np.sign(u) * (32768.0 / 255.0) * (256.0 ** (np.abs(u) / 128.0) - 1.0)
    assert np.allclose(lin, want, rtol=2e-6, atol=1e-3)
    # compression: monotone over the whole int16 range, clamps at both ends, and within one code of the textbook
    # formula (lin2ulaw uses a cubic log2 approximation, so it is not the exact logarithm)
    xs = np.arange(-40000, 40001, 7, dtype=np.float64)
    got = np.array([L.oracle_lin2ulaw(float(x)) for x in xs])
    assert np.all(np.diff(got) >= 0) and got[0] == 0 and got[-1] == 255
    ref = np.clip(128 + np.sign(xs) * 128 * np.log1p(255.0 * np.abs(xs) / 32768.0) / np.log(256.0), 0, 255)
    assert np.max(np.abs(got - np.floor(ref + 0.5))) <= 1
    assert L.oracle_lin2ulaw(0.0) == 128                                             # lpcnet_init(): last_exc


# ---------------------------------------------------------------------------------------------------------
# (c) activations: tansig table + second-order interpolation vs libm
# ---------------------------------------------------------------------------------------------------------
def test_tanh_and_sigmoid_approximations_against_libm(oracle):
    m = oracle.lpcnet_model(synthetic_blob(0))
    xs = np.linspace(-12, 12, 48001).astype(np.float32)
    t = np.array([oracle.lib.oracle_tanh_approx(m, float(x)) for x in xs], dtype=np.float64)
    s = np.array([oracle.lib.oracle_sigmoid_approx(m, float(x)) for x in xs], dtype=np.float64)
    assert np.max(np.abs(t - np.tanh(xs.astype(np.float64)))) < 5e-6          # measured 2.8e-6: table rounded to 6 decimals
    assert np.max(np.abs(s - 1 / (1 + np.exp(-xs.astype(np.float64))))) < 3e-6
    assert np.all(np.diff(t) >= -1e-6) and np.array_equal(t, -t[::-1])        # monotone up to the table's rounding; odd
    assert t[0] == -1.0 and t[-1] == 1.0                                       # saturates at |x| >= 8
    tab = oracle.lpcnet_table(m, 0, 201)
    assert np.array_equal(tab, np.round(np.tanh(0.04 * np.arange(201)), 6).astype(np.float32))


def test_sampling_threshold_is_a_uniform_draw_against_the_sigmoid(oracle):
    # sample_mdense compares `sampling_logit_table[r] < logit`; lpcnet_init() fills the table with
    # -log((1-p)/p), p = .025 + .95 r/255: that is exactly "p_r < sigmoid(logit)", a uniform draw on a 256-point grid
    m = oracle.lpcnet_model(synthetic_blob(0))
    tab = oracle.lpcnet_table(m, 1, 256).astype(np.float64)
    p = 0.025 + 0.95 * np.arange(256) / 255.0
    assert np.allclose(tab, np.log(p / (1 - p)), rtol=0, atol=2e-6) and np.all(np.diff(tab) > 0)
    rng = np.random.default_rng(3)
    logits = rng.normal(0, 3, 2000)
    sig = 1 / (1 + np.exp(-logits))
    for lg, sg in zip(logits[:200], sig[:200]):
        by_table = tab < np.float32(lg)
        by_prob = p < sg
        assert np.count_nonzero(by_table != by_prob) <= 1                     # equal except on a grid point itself


# ---------------------------------------------------------------------------------------------------------
# (d) Levinson-Durbin (_celt_lpc) on closed-form autocorrelations, and lpc_from_cepstrum vs numpy + scipy
# ---------------------------------------------------------------------------------------------------------
def test_levinson_recovers_known_predictors(oracle):
    # AR(1), x[n] = rho x[n-1] + e: r[k] = rho^k  ->  A(z) = 1 - rho z^-1, i.e. lpc = [-rho, 0, ...]
    for rho in (0.5, -0.3, 0.9):
        lpc, err = oracle.celt_lpc(rho ** np.arange(17), 16)
        assert abs(lpc[0] + rho) < 1e-6 and np.max(np.abs(lpc[1:])) < 1e-5
        assert abs(err - (1 - rho * rho)) < 1e-5
    # AR(2) with poles r e^{+-j theta}: a1 = -2 r cos(theta), a2 = r^2; autocorrelation from its Yule-Walker recursion
    r_, th = 0.8, 1.0
    a1, a2 = -2 * r_ * np.cos(th), r_ * r_
    ac = np.zeros(17)
    ac[0] = 1.0
    ac[1] = -a1 / (1 + a2)
    for k in range(2, 17):
        ac[k] = -a1 * ac[k - 1] - a2 * ac[k - 2]
    lpc, _ = oracle.celt_lpc(ac, 16)
    assert abs(lpc[0] - a1) < 1e-5 and abs(lpc[1] - a2) < 1e-5 and np.max(np.abs(lpc[2:])) < 1e-4
    # all-zero autocorrelation: the source leaves the predictor at zero
    lpc, err = oracle.celt_lpc(np.zeros(17), 16)
    assert not lpc.any() and err == 0.0


def _lpc_from_cepstrum_numpy(cep):
    """freq.c lpc_from_cepstrum from its DEFINITION, float64: idct-II of the Bark cepstrum -> 10^x band energies
    -> triangular interpolation onto 161 bins -> 320-point inverse real FFT -> noise floor, lag window -> Toeplitz."""
    from scipy.linalg import solve_toeplitz
    nb = 18
    eband = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40])
    comp = np.array([0.8, 1, 1, 1, 1, 1, 1, 1, 0.666667, 0.5, 0.5, 0.5, 0.333333, 0.25, 0.25, 0.2, 0.166667, 0.173913])
    c = np.array(cep, dtype=np.float64)
    c[0] += 4
    j = np.arange(nb)
    basis = np.cos((j[:, None] + 0.5) * j[None, :] * np.pi / nb)          # [band i][coef j]
    basis[:, 0] *= np.sqrt(0.5)
    ex = 10.0 ** ((basis @ c) * np.sqrt(2.0 / nb)) * comp
    xr = np.zeros(161)
    for i in range(nb - 1):
        n = (eband[i + 1] - eband[i]) * 4
        frac = np.arange(n) / n
        xr[eband[i] * 4: eband[i] * 4 + n] = (1 - frac) * ex[i] + frac * ex[i + 1]
    xr[160] = 0
    ac = (np.fft.irfft(xr, 320) * 320)[:17]                                  # opus_fft scales 1/N, inverse_transform xN
    ac[0] += ac[0] * 1e-4 + 320 / 12 / 38.0
    ac[1:] *= 1 - 6e-5 * np.arange(1, 17) ** 2
    return solve_toeplitz(ac[:16], -ac[1:17]), ac


def test_lpc_from_cepstrum_against_numpy_irfft_and_scipy_toeplitz(oracle):
    m = oracle.lpcnet_model(synthetic_blob(0))
    devs = []
    for seed in range(40):
        cep = synthetic_features(seed, 1)[0, :18]
        want, ac = _lpc_from_cepstrum_numpy(cep)
        # _celt_lpc stops early once the prediction gain passes 30 dB; these inputs stay below it (checked, not assumed)
        from scipy.linalg import solve_toeplitz
        err = ac[0]
        for order in range(1, 17):
            a = solve_toeplitz(ac[:order], -ac[1:order + 1])
            err = ac[0] + a @ ac[1:order + 1]
            assert err >= 0.001 * ac[0], "test input reaches the early exit of _celt_lpc"
        got = oracle.lpc_from_cepstrum(m, cep).astype(np.float64)
        devs.append(np.max(np.abs(got - want)))
        # the Levinson step alone, fed the float64 autocorrelation, is accurate to float32 rounding
        assert np.max(np.abs(oracle.celt_lpc(ac, 16)[0] - want)) < 1e-4
    # float32 chain (pow, interpolation, 160-term direct inverse DFT, Levinson) against float64 FFT + Toeplitz solve,
    # coefficients of magnitude <= ~3.  Stated tolerance: 90 % of the cases within 5e-4, every case within 1e-2 -- the
    # float32 rounding of the autocorrelation (relative ~2e-5 after the 160-term sum) is amplified by the Toeplitz
    # system's condition number (up to a few thousand with the -40 dB noise floor); a structural error (band edges,
    # compensation, lag window, FFT scaling) shows up as 0.1 .. 1.
    devs = np.sort(devs)
    assert devs[int(0.9 * len(devs)) - 1] < 5e-4 and devs[-1] < 1e-2, devs[-5:]


# ---------------------------------------------------------------------------------------------------------
# (e) the sample-rate network from its layer definitions, teacher-forced: all 255 node logits
# ---------------------------------------------------------------------------------------------------------
def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


class SampleNetNumpy:
    """run_sample_network as the Keras model defines it (lpcnet.py: GRU(384, reset_after) on the sum of three
    embeddings and the frame conditioning; GRU(16, reset_after) on its output plus conditioning; MDense(256) with two
    tanh channels).  float64 dense algebra with exact tanh / sigmoid -- nothing here follows the C text's loops."""

    def __init__(self, blob):
        d, w = unpack_blob(blob)
        na, nb = d.gru_a, d.gru_b
        self.na, self.nb = na, nb
        R = np.zeros((na, 3 * na))                      # recurrent kernel [input unit][gate*na + output unit]
        pos, blk = 0, 0
        idx = w["gru_a_idx"]
        for grp in range(3 * na // 8):
            cnt = idx[pos]; pos += 1
            for _ in range(cnt):
                col = idx[pos]; pos += 1
                R[col:col + 4, grp * 8:grp * 8 + 8] += w["gru_a_w"][blk].astype(np.float64)   # block [4 in][8 out]
                blk += 1
        for g in range(3):
            R[np.arange(na), g * na + np.arange(na)] += w["gru_a_diag"][g * na:(g + 1) * na]
        self.R, self.rb = R, w["gru_a_rbias"].astype(np.float64)
        self.E = [w[k].astype(np.float64) for k in ("embed_sig", "embed_pred", "embed_exc")]
        self.Wb_in, self.Wb_rec = w["gru_b_w_in"].astype(np.float64), w["gru_b_w_rec"].astype(np.float64)
        self.bb = w["gru_b_bias"].astype(np.float64)
        self.fc_w = w["dual_fc_w"].astype(np.float64)                     # [node][channel][input]
        self.fc_b = w["dual_fc_bias"].astype(np.float64).reshape(2, -1)
        self.fc_f = w["dual_fc_factor"].astype(np.float64).reshape(2, -1)

    def step(self, ha, hb, cond_a, cond_b, si, pi, ei):
        na, nb = self.na, self.nb
        x = cond_a + self.E[0][si] + self.E[1][pi] + self.E[2][ei]
        rec = ha @ self.R + self.rb
        z = _sig(x[:na] + rec[:na]); r = _sig(x[na:2 * na] + rec[na:2 * na])
        hh = np.tanh(x[2 * na:] + r * rec[2 * na:])
        ha2 = z * ha + (1 - z) * hh
        gi = ha2 @ self.Wb_in + self.bb[0] + cond_b
        gr = hb @ self.Wb_rec + self.bb[1]
        zb = _sig(gi[:nb] + gr[:nb]); rb = _sig(gi[nb:2 * nb] + gr[nb:2 * nb])
        hhb = np.tanh(gi[2 * nb:] + rb * gr[2 * nb:])
        hb2 = zb * hb + (1 - zb) * hhb
        logits = sum(self.fc_f[c] * np.tanh(self.fc_b[c] + self.fc_w[:, c, :] @ hb2) for c in range(2))
        return ha2, hb2, logits


@pytest.mark.parametrize("order", [GRUA_INPUT_FIRST, GRUA_RECUR_FIRST])
def test_sample_step_teacher_forced_against_layer_definitions(oracle, order):
    blob = synthetic_blob(0, gru_a_order=order)
    m = oracle.lpcnet_model(blob)
    net = SampleNetNumpy(blob)
    rng = np.random.default_rng(11)
    dec = oracle.decoder(m)
    worst_logit = worst_a = worst_b = 0.0
    for trial in range(24):
        ha = rng.uniform(-0.9, 0.9, 384).astype(np.float32)
        hb = rng.uniform(-0.9, 0.9, 16).astype(np.float32)
        ca = rng.normal(0, 0.5, 1152).astype(np.float32)
        cb = rng.normal(0, 0.5, 48).astype(np.float32)
        si, pi, ei = (int(v) for v in rng.integers(0, 256, 3))
        dec.set_state(ha, hb, ca, cb)
        got = dec.sample_step(ei, si, pi)                                   # (last_exc, last_sig_ulaw, pred_ulaw)
        ha2, hb2, want = net.step(ha.astype(np.float64), hb.astype(np.float64), ca.astype(np.float64),
                                  cb.astype(np.float64), si, pi, ei)
        worst_a = max(worst_a, np.max(np.abs(dec.tap(3, 384) - ha2)))
        worst_b = max(worst_b, np.max(np.abs(dec.tap(4, 16) - hb2)))
        worst_logit = max(worst_logit, np.max(np.abs(got[1:] - want[1:])))
    # Stated tolerances (float32 sequential sums + table activations vs float64 dense algebra + exact activations):
    # GRU states 1e-4, node logits 2e-3 (logit scale ~ +-5; sixteen 1e-4 state errors through |w| ~ 1 and factor ~ 1.5-3.5)
    assert worst_a < 1e-4 and worst_b < 1e-4, (worst_a, worst_b)
    assert worst_logit < 2e-3, worst_logit


def test_gru_a_association_order_is_a_model_flag(oracle):
    rng = np.random.default_rng(5)
    ha = rng.uniform(-0.9, 0.9, 384).astype(np.float32)
    hb = rng.uniform(-0.9, 0.9, 16).astype(np.float32)
    ca = rng.normal(0, 0.5, 1152).astype(np.float32)
    cb = rng.normal(0, 0.5, 48).astype(np.float32)
    states = []
    for order in (GRUA_INPUT_FIRST, GRUA_RECUR_FIRST):
        dec = oracle.decoder(oracle.lpcnet_model(synthetic_blob(0, gru_a_order=order)))
        dec.set_state(ha, hb, ca, cb)
        dec.sample_step(128, 100, 140)
        states.append(dec.tap(3, 384))
    assert not np.array_equal(states[0], states[1])                 # a different float association ...
    assert np.max(np.abs(states[0] - states[1])) < 1e-5             # ... of the same real-number expression
    # (a 1e-7 difference flips one of the sampler's 8 comparisons per sample against a 256-level threshold grid only
    # about once per 10^5..10^6 samples: 4 s of synthetic-model audio came out identical under both orders, so a
    # waveform-level assertion would be a coin toss; the state-level one above is deterministic)
    bad = bytearray(synthetic_blob(0)); bad[8 + 4 * 15] = 7
    with pytest.raises(ValueError):
        oracle.lpcnet_model(bytes(bad))


# ---------------------------------------------------------------------------------------------------------
# (f) frame-rate network from its layer definitions (causal Conv1D k=3 x2, Dense x2, two projections), with the
#     two-frame look-ahead of lpcnet.c
# ---------------------------------------------------------------------------------------------------------
def test_frame_network_against_layer_definitions(oracle):
    blob = synthetic_blob(0)
    d, w = unpack_blob(blob)
    W = {k: v.astype(np.float64) for k, v in w.items() if v.dtype == np.float32}
    m = oracle.lpcnet_model(blob)
    dec = oracle.decoder(m)
    F = 7
    feats = synthetic_features(21, F)
    pitch = np.clip(np.floor(0.1 + 50 * feats[:, 18].astype(np.float64) + 100), 33, 255).astype(int)
    x = np.concatenate([feats.astype(np.float64), W["embed_pitch"][pitch]], axis=1)          # (F, 84)

    def conv(xpad, wk, b):          # xpad: (F+2, C) with two leading zero rows; kernel rows [oldest | mid | newest]
        c = xpad.shape[1]
        return np.tanh(b + xpad[:-2] @ wk[:c] + xpad[1:-1] @ wk[c:2 * c] + xpad[2:] @ wk[2 * c:])

    c1 = conv(np.vstack([np.zeros((2, 84)), x]), W["conv1_w"], W["conv1_b"])
    c1[:1] = 0                                   # lpcnet.c: frame_count < FEATURE_CONV1_DELAY -> conv1 output zeroed
    c2 = conv(np.vstack([np.zeros((2, 128)), c1]), W["conv2_w"], W["conv2_b"])
    c2[:2] = 0                                   # frame_count < FEATURES_DELAY -> conv2 output zeroed
    d1 = np.tanh(W["dense1_b"] + c2 @ W["dense1_w"])
    cond = np.tanh(W["dense2_b"] + d1 @ W["dense2_w"])
    want_a = W["gru_a_dense_b"] + cond @ W["gru_a_dense_w"]
    want_b = W["gru_b_dense_b"] + cond @ W["gru_b_dense_w"]
    worst = 0.0
    for f in range(F):
        dec.frame_network(feats[f])
        worst = max(worst, np.max(np.abs(dec.tap(0, 1152) - want_a[f])), np.max(np.abs(dec.tap(1, 48) - want_b[f])))
        # the LPC in force during frame f are those of frame f-2 (old_lpc delay line)
        if f >= 2:
            assert np.array_equal(dec.tap(2, 16), oracle.lpc_from_cepstrum(m, feats[f - 2, :18]))
        else:
            assert not dec.tap(2, 16).any()
    assert worst < 2e-4, worst                    # stated: float32 sequential sums + table tanh vs float64 + exact tanh
    # (the pitch index floor(.1 + 50 f18 + 100) clamped to [33, 255] is part of `x` above: a wrong index picks another
    # embedding row, std 0.5 per element, and the comparison fails by orders of magnitude)
    assert pitch.min() >= 33 and pitch.max() <= 255 and len(set(pitch.tolist())) > 3


# ---------------------------------------------------------------------------------------------------------
# (g) synthesis filter, de-emphasis, clamp and rounding: an independent float32 recursion driven by the forced
#     excitation must reproduce the oracle's int16 output exactly
# ---------------------------------------------------------------------------------------------------------
def test_forced_excitation_through_an_independent_synthesis_filter(oracle):
    blob = synthetic_blob(0)
    m = oracle.lpcnet_model(blob)
    F = 6
    feats = synthetic_features(33, F)
    n = (F - 2) * 160
    rng = np.random.default_rng(9)
    exc = np.clip(np.rint(128 + rng.normal(0, 25, n)), 0, 255).astype(np.uint8)
    exc[:40] = 128                                                # zero excitation first: output must stay silent
    dec = oracle.decoder(m, trace_cap=n)
    dec.force(exc, want_logits=False)
    pcm = np.concatenate([dec.synthesize(feats[f]) for f in range(F)])
    assert np.array_equal(dec.trace_exc, exc)
    u2l = oracle.lpcnet_table(m, 2, 256)
    f32 = np.float32
    sig = np.zeros(16, f32)
    mem = f32(0)
    out = np.zeros(F * 160, np.int16)
    for f in range(2, F):
        lpc = oracle.lpc_from_cepstrum(m, feats[f - 2, :18])      # LPC in force (pinned separately above)
        for i in range(160):
            pred = f32(0)
            for j in range(16):
                pred = f32(pred - f32(sig[j] * lpc[j]))
            s = f32(pred + u2l[exc[(f - 2) * 160 + i]])
            sig[1:] = sig[:-1].copy(); sig[0] = s
            y = f32(s + f32(f32(0.85) * mem))
            mem = y
            out[f * 160 + i] = int(np.floor(0.5 + float(min(max(y, f32(-32767)), f32(32767)))))
    assert not pcm[:320 + 40].any()
    assert np.array_equal(pcm, out)


# ---------------------------------------------------------------------------------------------------------
# (h) the slot that closes the open claim: vectors produced by xiph's own lpcnet_demo, when someone supplies them
# ---------------------------------------------------------------------------------------------------------
XIPH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lpcnet_xiph.npz")


@pytest.mark.skipif(not os.path.exists(XIPH), reason="tests/golden/lpcnet_xiph.npz not supplied: xiph/LPCNet sources and "
                    "nnet_data.c are absent from the reference tree (see tests/golden/README.md for how to make it)")
def test_oracle_against_real_xiph_vectors(oracle):
    g = np.load(XIPH)
    if "predefined_macros" in g.files:      # tests/golden/README.md: the xiph build's `gcc -dM -E` output travels with the vector
        macros = str(g["predefined_macros"])
        assert "DOT_PROD" not in macros, ("this vector comes from a build that compiles nnet.c's DOT_PROD (int8 qweight, "
                                          "subias) path -- a different arithmetic from the float path the oracle and the "
                                          "kernels implement (DESIGN.md 2); the int8 arrays are in <blob>.dotprod.npz")
    m = oracle.lpcnet_model(g["blob"].tobytes())
    got = oracle.lpcnet_utterance(m, g["features"][:, :20]).astype(np.int32)
    assert np.max(np.abs(got - g["pcm"].astype(np.int32))) <= 1           # north_star: +-1 LSB on 16-bit PCM
