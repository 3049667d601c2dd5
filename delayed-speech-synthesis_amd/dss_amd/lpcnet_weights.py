"""LPCNet weight blobs: layout (include/dss_lpcnet_blob.h), seeded synthetic generator, (un)packing.

The reference compiles xiph/LPCNet's trained weights into the extension as ``src/nnet_data.c``
(extensions/lpcnet/setup.py:34-36); that file is downloaded by xiph's autogen.sh and does not exist
in the reference tree, so this build treats weights as data.  ``make_synthetic_weights`` produces a
random-but-well-conditioned model of the published architecture (384-unit block-sparse GRU A at
5 % / 5 % / 20 % gate density in 8x4 blocks, 16-unit GRU B, 256-way dual-FC tree) so the path can be
exercised and timed; ``pack_blob`` writes the flat layout both the HIP library and the test oracle
parse.  A converter from a real ``nnet_data.c`` only has to fill the same dict.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Dict

import numpy as np

MAGIC = b"DSSLPCN1"


@dataclass(frozen=True)
class LPCNetDims:
    nb_features: int = 20
    nb_bands: int = 18
    embed_pitch_dim: int = 64
    pitch_max: int = 256
    conv1_out: int = 128
    conv2_out: int = 128
    dense1_out: int = 128
    dense2_out: int = 128
    gru_a: int = 384
    gru_b: int = 16
    dual_fc_out: int = 256
    lpc_order: int = 16


# (name, dtype) in blob order; shapes are derived from dims / the dict itself
_SECTIONS = [
    "embed_pitch", "conv1_w", "conv1_b", "conv2_w", "conv2_b", "dense1_w", "dense1_b", "dense2_w", "dense2_b",
    "gru_a_dense_w", "gru_a_dense_b", "gru_b_dense_w", "gru_b_dense_b", "embed_sig", "embed_pred", "embed_exc",
    "gru_a_rbias", "gru_a_diag", "gru_a_idx", "gru_a_w", "gru_b_bias", "gru_b_w_in", "gru_b_w_rec",
    "dual_fc_bias", "dual_fc_w", "dual_fc_factor",
]


def _shapes(d: LPCNetDims, nblocks: int, idx_len: int) -> Dict[str, tuple]:
    fin = d.nb_features + d.embed_pitch_dim
    na3, nb3 = 3 * d.gru_a, 3 * d.gru_b
    return {
        "embed_pitch": (d.pitch_max, d.embed_pitch_dim),
        "conv1_w": (3 * fin, d.conv1_out), "conv1_b": (d.conv1_out,),
        "conv2_w": (3 * d.conv1_out, d.conv2_out), "conv2_b": (d.conv2_out,),
        "dense1_w": (d.conv2_out, d.dense1_out), "dense1_b": (d.dense1_out,),
        "dense2_w": (d.dense1_out, d.dense2_out), "dense2_b": (d.dense2_out,),
        "gru_a_dense_w": (d.dense2_out, na3), "gru_a_dense_b": (na3,),
        "gru_b_dense_w": (d.dense2_out, nb3), "gru_b_dense_b": (nb3,),
        "embed_sig": (256, na3), "embed_pred": (256, na3), "embed_exc": (256, na3),
        "gru_a_rbias": (na3,), "gru_a_diag": (na3,),
        "gru_a_idx": (idx_len,), "gru_a_w": (nblocks, 4, 8),
        "gru_b_bias": (2, nb3), "gru_b_w_in": (d.gru_a, nb3), "gru_b_w_rec": (d.gru_b, nb3),
        "dual_fc_bias": (2 * d.dual_fc_out,), "dual_fc_w": (d.dual_fc_out, 2, d.gru_b),
        "dual_fc_factor": (2 * d.dual_fc_out,),
    }


GRUA_INPUT_FIRST, GRUA_RECUR_FIRST = 0, 1      # dss_blob_header.gru_a_order


def make_synthetic_weights(seed: int = 0, dims: LPCNetDims = LPCNetDims(),
                           density=(0.05, 0.05, 0.20), skew: float = 0.0) -> Dict[str, np.ndarray]:
    """Seeded random model of the published LPCNet decoder architecture.

    ``skew`` = 0 places the kept 8x4 blocks uniformly at random (every row group has about the same count).
    ``skew`` > 0 imitates what xiph's training does (magnitude pruning keeps the top-k blocks of a gate
    GLOBALLY, so some row groups keep many more blocks than others): each row group draws a log-normal
    "importance" with that sigma and the kept blocks are the top-k of importance x per-block noise."""
    d = dims
    rng = np.random.default_rng(seed)
    f32 = np.float32
    fin = d.nb_features + d.embed_pitch_dim
    na, nb = d.gru_a, d.gru_b

    def normal(shape, std):
        return (rng.standard_normal(shape) * std).astype(f32)

    w: Dict[str, np.ndarray] = {}
    w["embed_pitch"] = normal((d.pitch_max, d.embed_pitch_dim), 0.5)
    w["conv1_w"] = normal((3 * fin, d.conv1_out), 1.2 / np.sqrt(3 * fin))
    w["conv1_b"] = normal((d.conv1_out,), 0.1)
    w["conv2_w"] = normal((3 * d.conv1_out, d.conv2_out), 1.5 / np.sqrt(3 * d.conv1_out))
    w["conv2_b"] = normal((d.conv2_out,), 0.1)
    w["dense1_w"] = normal((d.conv2_out, d.dense1_out), 1.5 / np.sqrt(d.conv2_out))
    w["dense1_b"] = normal((d.dense1_out,), 0.1)
    w["dense2_w"] = normal((d.dense1_out, d.dense2_out), 1.5 / np.sqrt(d.dense1_out))
    w["dense2_b"] = normal((d.dense2_out,), 0.1)
    w["gru_a_dense_w"] = normal((d.dense2_out, 3 * na), 0.08)
    w["gru_a_dense_b"] = normal((3 * na,), 0.1)
    w["gru_b_dense_w"] = normal((d.dense2_out, 3 * nb), 0.08)
    w["gru_b_dense_b"] = normal((3 * nb,), 0.1)
    for name in ("embed_sig", "embed_pred", "embed_exc"):
        w[name] = normal((256, 3 * na), 0.4)
    w["gru_a_rbias"] = normal((3 * na,), 0.1)
    w["gru_a_diag"] = normal((3 * na,), 0.3)

    # block-sparse recurrent matrix: per gate, a fixed number of 8(out) x 4(in) blocks chosen at random
    groups_per_gate, col_blocks = na // 8, na // 4
    idx = []
    blocks = []
    for g, dens in enumerate(density):
        n_keep = int(round(dens * groups_per_gate * col_blocks))
        if skew > 0:
            score = np.exp(skew * rng.standard_normal((groups_per_gate, 1))) * rng.random((groups_per_gate, col_blocks))
            keep = np.argsort(-score.ravel(), kind="stable")[:n_keep]
        else:
            keep = rng.choice(groups_per_gate * col_blocks, size=n_keep, replace=False)
        mask = np.zeros(groups_per_gate * col_blocks, dtype=bool)
        mask[keep] = True
        mask = mask.reshape(groups_per_gate, col_blocks)
        std = 0.25 if g < 2 else 0.2
        for grp in range(groups_per_gate):
            cols = np.nonzero(mask[grp])[0]
            idx.append(len(cols))
            idx.extend((cols * 4).tolist())
            for _ in cols:
                blocks.append(normal((4, 8), std))
    w["gru_a_idx"] = np.asarray(idx, dtype=np.int32)
    w["gru_a_w"] = np.stack(blocks).astype(f32)

    w["gru_b_bias"] = normal((2, 3 * nb), 0.1)
    w["gru_b_w_in"] = normal((na, 3 * nb), 0.1)
    w["gru_b_w_rec"] = normal((nb, 3 * nb), 0.3)

    # dual FC tree: channel 0 carries a fixed preference that concentrates the sampled mu-law value
    # around 128 (small excitation), channel 1 the state-dependent part.
    nout = d.dual_fc_out
    bias = np.zeros(2 * nout, f32)
    fac = np.zeros(2 * nout, f32)
    wfc = np.zeros((nout, 2, nb), f32)
    wfc[:, 0, :] = normal((nout, nb), 0.05)
    wfc[:, 1, :] = normal((nout, nb), 1.0)
    bias[:nout] = 3.0
    bias[nout:] = normal((nout,), 0.3)
    fac[nout:] = normal((nout,), 1.5)
    pref = [0.0, 3.5, 3.0, 2.0, 1.0, 0.3, 0.0, 0.0]          # |logit offset| per tree level
    for node in range(1, nout):
        level = node.bit_length() - 1                          # node = (1 << level) | prefix
        if level == 0:
            continue
        msb = (node >> (level - 1)) & 1                        # first decided bit of the prefix
        fac[node] = (-pref[level] if msb else pref[level])
    fac[:nout] += normal((nout,), 0.2)
    w["dual_fc_bias"], w["dual_fc_w"], w["dual_fc_factor"] = bias, wfc, fac
    return w


def pack_blob(w: Dict[str, np.ndarray], dims: LPCNetDims = LPCNetDims(), gru_a_order: int = GRUA_INPUT_FIRST,
              source_branches: int = 0) -> bytes:
    d = dims
    if gru_a_order not in (GRUA_INPUT_FIRST, GRUA_RECUR_FIRST):
        raise ValueError("gru_a_order must be 0 (input first, xiph 2021) or 1 (recurrent first, xiph 2019-2020)")
    nblocks = int(w["gru_a_w"].shape[0])
    idx_len = int(w["gru_a_idx"].shape[0])
    shapes = _shapes(d, nblocks, idx_len)
    header = struct.pack(
        "<8s22i", MAGIC, 1, d.nb_features, d.nb_bands, d.embed_pitch_dim, d.pitch_max, d.conv1_out, d.conv2_out,
        d.dense1_out, d.dense2_out, d.gru_a, d.gru_b, d.dual_fc_out, d.lpc_order, nblocks, idx_len, int(gru_a_order),
        int(source_branches), 0, 0, 0, 0, 0)
    assert len(header) == 96
    parts = [header]
    for name in _SECTIONS:
        a = np.ascontiguousarray(w[name])
        if tuple(a.shape) != shapes[name]:
            raise ValueError(f"{name}: shape {a.shape} != expected {shapes[name]}")
        want = np.int32 if name == "gru_a_idx" else np.float32
        if a.dtype != want:
            raise ValueError(f"{name}: dtype {a.dtype} != {want}")
        parts.append(a.tobytes())
    return b"".join(parts)


def blob_source_branches(blob: bytes) -> int:
    """dss_blob_header.source_branches: 0 unknown (synthetic / older blobs), bit 0 float branch, bit 1 DOT_PROD branch present
    in the nnet_data.c the blob was converted from."""
    return int(struct.unpack_from("<i", blob, 8 + 4 * 16)[0])


def unpack_blob(blob: bytes):
    vals = struct.unpack_from("<8s15i", blob, 0)
    if vals[0] != MAGIC or vals[1] != 1:
        raise ValueError("not a DSSLPCN1 blob")
    (nf, nbands, epd, pmax, c1, c2, d1, d2, na, nb, nout, order, nblocks, idx_len) = vals[2:16]
    dims = LPCNetDims(nf, nbands, epd, pmax, c1, c2, d1, d2, na, nb, nout, order)
    shapes = _shapes(dims, nblocks, idx_len)
    off = 96
    w = {}
    for name in _SECTIONS:
        dt = np.int32 if name == "gru_a_idx" else np.float32
        n = int(np.prod(shapes[name]))
        w[name] = np.frombuffer(blob, dtype=dt, count=n, offset=off).reshape(shapes[name]).copy()
        off += 4 * n
    if off != len(blob):
        raise ValueError("blob length mismatch")
    return dims, w


def blob_gru_a_order(blob: bytes) -> int:
    return struct.unpack_from("<i", blob, 8 + 4 * 15)[0]


_cache: Dict[tuple, bytes] = {}


def synthetic_blob(seed: int = 0, gru_a_order: int = GRUA_INPUT_FIRST, skew: float = 0.0) -> bytes:
    """Blob of a synthetic model (cached per process).  TESTS AND BENCHMARKS ONLY: random weights make noise,
    not speech."""
    key = (seed, gru_a_order, skew)
    if key not in _cache:
        _cache[key] = pack_blob(make_synthetic_weights(seed, skew=skew), gru_a_order=gru_a_order)
    return _cache[key]


def algorithmic_bytes_per_sample(blob: bytes) -> float:
    """weights touched once per output sample at fp32, plus I/O."""
    dims, w = unpack_blob(blob)
    na, nb = dims.gru_a, dims.gru_b
    nnz = int(w["gru_a_w"].size)
    floats = 3 * (3 * na) + nnz + 3 * na + 3 * nb * (na + nb) + 2 * nb * 8 + 16
    return 4.0 * floats + 2.0 + 80.0 / 160.0


def synthetic_features(seed: int, n_frames: int = 100) -> np.ndarray:
    """SURVEY.md section 8(d) config 1/2 input: (n_frames, 20) float32, seeded per utterance."""
    rng = np.random.default_rng(seed)
    scale = np.array([4, 2, 1, 1, .8, .7, .6, .5, .5, .4, .4, .3, .3, .3, .2, .2, .2, .2], dtype=np.float64)
    f = np.empty((n_frames, 20), dtype=np.float32)
    f[:, :18] = (rng.standard_normal((n_frames, 18)) * scale).astype(np.float32)
    f[:, 18] = rng.uniform(-1.4, 3.1, n_frames).astype(np.float32)
    f[:, 19] = rng.uniform(-0.5, 0.5, n_frames).astype(np.float32)
    return f
