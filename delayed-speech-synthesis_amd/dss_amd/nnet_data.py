"""Converter from xiph/LPCNet's generated ``src/nnet_data.c`` to the weight blob of include/dss_lpcnet_blob.h.

The reference compiles that file into its extension (extensions/lpcnet/setup.py:34-36); xiph's ``autogen.sh`` downloads
it, so it exists neither in the reference tree nor in this image.  This module is what a maintainer who has the file
runs once::

    python -m dss_amd.nnet_data path/to/LPCNet/src/nnet_data.c lpcnet.blob
    DSS_LPCNET_WEIGHTS=lpcnet.blob python decode_online.py ...

**Status: [UNVERIFIED] against a real nnet_data.c.**  The array names and layouts below are those of xiph's
``dump_lpcnet.py`` output as of late 2021 (the era of the reference's translation-unit list), written down from the
published sources; every array is checked for its exact size against the architecture constants, so a different
revision fails loudly instead of producing a wrong model.  A name that differs can be remapped with ``names={...}``.
What is tested here (tests/test_cpu_nnet_data.py) is the parser and the mapping on a synthetic file written in the same
C syntax, including the ``#ifdef DOT_PROD`` / ``#else`` pairs xiph emits for quantisable layers.

**Which arithmetic does the reference's build use?  Unverifiable here, so BOTH branches are kept.**  The blob (what the
kernels and the oracle run on) is built from the ``#else`` branch: float weights, the generic ``vec.h`` path, which is
what a plain ``-O2`` x86-64 build selects in the xiph revisions this was written from (no ``-mavx*`` reaches the compiler,
extensions/lpcnet/setup.py:27-29,48-52).  Later public revisions select ``vec_avx.h`` whenever ``__SSE2__`` is defined
(every x86-64 build) and define ``DOT_PROD`` there: int8 ``qweight`` blocks, a quantised GRU input, ``subias`` and
vectorised exp-based activations -- a different arithmetic.  The converter therefore also keeps the ``#ifdef DOT_PROD``
branch's arrays (int8 weights and every array that exists only there: scales, ``subias``) in a side file next to the blob
(``<out>.dotprod.npz``) and records in the blob header (``source_branches``) which branches the source had.  No kernel
reads the side file: whether one has to is decided by the first xiph-produced vector (tests/golden/README.md), whose
recipe stores the build's predefined macros beside it.

Layout facts relied on (xiph src/nnet.c, generic path):
  * DenseLayer / conv1d ``input_weights``: input-major, ``w[i * nb_neurons + j]`` (sgemv_accum with stride = outputs);
    conv1d inputs are [oldest frame | ... | newest frame].
  * EmbeddingLayer: one row of ``dim`` floats per index.
  * gru_a (sparse): ``recurrent_weights`` as 8x4 blocks ``[4 inputs][8 rows]`` in ``idx`` order, ``idx`` = per group of
    8 rows a count followed by the first input column of each block; ``diag_weights[3N]``; ``bias[2 * 3N]`` of which
    the second half (recurrent bias) is what compute_sparse_gru reads.
  * gru_b: ``bias[2 * 3N]`` (input, recurrent), ``input_weights[384][48]``, ``recurrent_weights[16][48]``, both
    input-major.
  * dual_fc (MDenseLayer as sampled by sample_mdense): ``bias[2 * 256]``, ``factor[2 * 256]`` and
    ``input_weights[256][2][16]`` (node, channel, input).
"""
from __future__ import annotations

import re
import sys
from typing import Dict, Optional

import numpy as np

from .lpcnet_weights import LPCNetDims, pack_blob

# blob key -> (C array name, dtype)
DEFAULT_NAMES = {
    "embed_pitch": "embed_pitch_weights",
    "conv1_w": "feature_conv1_weights", "conv1_b": "feature_conv1_bias",
    "conv2_w": "feature_conv2_weights", "conv2_b": "feature_conv2_bias",
    "dense1_w": "feature_dense1_weights", "dense1_b": "feature_dense1_bias",
    "dense2_w": "feature_dense2_weights", "dense2_b": "feature_dense2_bias",
    "gru_a_dense_w": "gru_a_dense_feature_weights", "gru_a_dense_b": "gru_a_dense_feature_bias",
    "gru_b_dense_w": "gru_b_dense_feature_weights", "gru_b_dense_b": "gru_b_dense_feature_bias",
    "embed_sig": "gru_a_embed_sig_weights", "embed_pred": "gru_a_embed_pred_weights",
    "embed_exc": "gru_a_embed_exc_weights",
    "gru_a_bias": "gru_a_bias", "gru_a_diag": "gru_a_recurrent_weights_diag",
    "gru_a_idx": "gru_a_recurrent_weights_idx", "gru_a_w": "gru_a_recurrent_weights",
    "gru_b_bias": "gru_b_bias", "gru_b_w_in": "gru_b_weights", "gru_b_w_rec": "gru_b_recurrent_weights",
    "dual_fc_bias": "dual_fc_bias", "dual_fc_w": "dual_fc_weights", "dual_fc_factor": "dual_fc_factor",
}

_ARRAY = re.compile(r"(?:static\s+)?const\s+(float|int|qweight|opus_int8|signed\s+char)\s+(\w+)\s*\[[^\]]*\]\s*=\s*\{(.*?)\}\s*;",
                    re.S)


BRANCH_FLOAT, BRANCH_DOT_PROD = 1, 2          # bits of dss_blob_header.source_branches


def _select_branch(text: str, dot_prod: bool):
    """Resolve every ``#ifdef DOT_PROD`` / ``#else`` / ``#endif`` block to one branch (``dot_prod``: the ``#ifdef`` side,
    else the ``#else`` side) and drop other preprocessor lines.  Returns (text, number of DOT_PROD blocks seen)."""
    out, stack, n_blocks = [], [], 0            # stack of [is_dot_prod_block, currently_in_else]
    for line in text.splitlines():
        t = line.strip()
        if t.startswith("#if"):
            is_dp = bool(re.match(r"#\s*ifdef\s+DOT_PROD\b", t))
            n_blocks += is_dp
            stack.append([is_dp, False])
            continue
        if t.startswith("#else"):
            if stack:
                stack[-1][1] = True
            continue
        if t.startswith("#endif"):
            if stack:
                stack.pop()
            continue
        if t.startswith("#"):
            continue
        if any(is_dp and (in_else == dot_prod) for is_dp, in_else in stack):
            continue                # the other branch
        out.append(line)
    return "\n".join(out), n_blocks


def _strip_dot_prod(text: str) -> str:
    """Keep the ``#else`` (float) branch of every ``#ifdef DOT_PROD`` block and drop other preprocessor lines."""
    return _select_branch(text, dot_prod=False)[0]


def _c_int(tok: str) -> int:
    """A C integer literal as nnet_data.c prints them: hexadecimal with 0x, everything else DECIMAL (a leading zero is
    padding there, not an octal prefix -- and int(t, 0) refuses '010')."""
    t = tok.lstrip("+-")
    v = int(t, 16) if t[:2].lower() == "0x" else int(t, 10)
    return -v if tok.startswith("-") else v


def _parse_resolved(text: str, dot_prod: bool = False) -> Dict[str, np.ndarray]:
    """``dot_prod``: which side of the ``#ifdef DOT_PROD`` pairs `text` was resolved to.  It decides the element type of
    the ``qweight`` arrays (typedef'd to ``float`` in the ``#else`` build, ``signed char`` under DOT_PROD) -- NOT the look of
    their tokens: a float array whose values all print without a decimal point stays float32."""
    arrays: Dict[str, np.ndarray] = {}
    for ctype, name, body in _ARRAY.findall(text):
        toks = [t for t in re.split(r"[\s,]+", body.strip()) if t]
        ctype = " ".join(ctype.split())
        if ctype == "int":
            arrays[name] = np.array([_c_int(t) for t in toks], dtype=np.int32)
        elif ctype in ("opus_int8", "signed char") or (ctype == "qweight" and dot_prod):
            vals = [_c_int(t) for t in toks]
            bad = [v for v in vals if not -128 <= v <= 127]
            if bad:
                raise ValueError(f"{name}: int8 array holds {bad[0]} (outside -128..127)")
            arrays[name] = np.array(vals, dtype=np.int8)
        else:
            arrays[name] = np.array([float(t.rstrip("fF")) for t in toks], dtype=np.float32)
    return arrays


def _uncomment(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def parse_c_arrays(text: str) -> Dict[str, np.ndarray]:
    """All ``const float/int NAME[...] = {...};`` initialisers of a C file -> {name: 1-D array}, ``#else`` (float) branch
    of the ``#ifdef DOT_PROD`` pairs."""
    return _parse_resolved(_strip_dot_prod(_uncomment(text)))


def parse_c_arrays_both(text: str):
    """Both resolutions of the ``#ifdef DOT_PROD`` pairs: (float-branch arrays, DOT_PROD-only arrays, branches).
    The second dict holds every array of the ``#ifdef`` side that is absent from or different in the ``#else`` side: the
    int8 weight arrays (same names as their float counterparts) and whatever exists only under DOT_PROD (scales,
    ``subias``).  ``branches``: BRANCH_FLOAT | BRANCH_DOT_PROD when the file has such pairs, BRANCH_FLOAT otherwise."""
    text = _uncomment(text)
    ftext, n_blocks = _select_branch(text, dot_prod=False)
    fl = _parse_resolved(ftext)
    if not n_blocks:
        return fl, {}, BRANCH_FLOAT
    dp_all = _parse_resolved(_select_branch(text, dot_prod=True)[0], dot_prod=True)
    dp = {k: v for k, v in dp_all.items() if k not in fl or v.dtype != fl[k].dtype or v.size != fl[k].size or not np.array_equal(v, fl[k])}
    return fl, dp, BRANCH_FLOAT | BRANCH_DOT_PROD


def weights_from_nnet_data(text: str, dims: LPCNetDims = LPCNetDims(), names: Optional[Dict[str, str]] = None):
    """C source text of nnet_data.c -> the dict ``pack_blob`` takes.  Raises ValueError naming the array on any size
    mismatch."""
    nm = dict(DEFAULT_NAMES)
    nm.update(names or {})
    arrays = parse_c_arrays(text)               # the #else (float) branch: the arithmetic the kernels implement
    d = dims
    fin = d.nb_features + d.embed_pitch_dim
    na, nb = d.gru_a, d.gru_b

    def take(key, shape, dtype=np.float32):
        cname = nm[key]
        if cname not in arrays:
            raise ValueError(f"array '{cname}' (for {key}) not found; arrays present: {sorted(arrays)[:8]} ...")
        a = arrays[cname]
        n = int(np.prod(shape))
        if a.size != n:
            raise ValueError(f"array '{cname}' has {a.size} elements, the architecture needs {n} {shape}")
        return np.ascontiguousarray(a.astype(dtype).reshape(shape))

    w: Dict[str, np.ndarray] = {}
    w["embed_pitch"] = take("embed_pitch", (d.pitch_max, d.embed_pitch_dim))
    w["conv1_w"] = take("conv1_w", (3 * fin, d.conv1_out)); w["conv1_b"] = take("conv1_b", (d.conv1_out,))
    w["conv2_w"] = take("conv2_w", (3 * d.conv1_out, d.conv2_out)); w["conv2_b"] = take("conv2_b", (d.conv2_out,))
    w["dense1_w"] = take("dense1_w", (d.conv2_out, d.dense1_out)); w["dense1_b"] = take("dense1_b", (d.dense1_out,))
    w["dense2_w"] = take("dense2_w", (d.dense1_out, d.dense2_out)); w["dense2_b"] = take("dense2_b", (d.dense2_out,))
    w["gru_a_dense_w"] = take("gru_a_dense_w", (d.dense2_out, 3 * na)); w["gru_a_dense_b"] = take("gru_a_dense_b", (3 * na,))
    w["gru_b_dense_w"] = take("gru_b_dense_w", (d.dense2_out, 3 * nb)); w["gru_b_dense_b"] = take("gru_b_dense_b", (3 * nb,))
    for key in ("embed_sig", "embed_pred", "embed_exc"):
        w[key] = take(key, (256, 3 * na))
    w["gru_a_rbias"] = take("gru_a_bias", (2, 3 * na))[1].copy()          # recurrent half (compute_sparse_gru)
    w["gru_a_diag"] = take("gru_a_diag", (3 * na,))
    cname = nm["gru_a_idx"]
    if cname not in arrays:
        raise ValueError(f"array '{cname}' (for gru_a_idx) not found")
    idx = arrays[cname].astype(np.int32)
    # walk the index list: 3 gates x N/8 row groups, each "count, positions..."
    pos, nblocks = 0, 0
    for _ in range(3 * na // 8):
        if pos >= idx.size:
            raise ValueError(f"array '{cname}' ends after {pos} entries: fewer than {3 * na // 8} row groups")
        c = int(idx[pos])
        if c < 0 or pos + 1 + c > idx.size or (idx[pos + 1:pos + 1 + c] % 4).any() or (idx[pos + 1:pos + 1 + c] >= na).any():
            raise ValueError(f"array '{cname}': malformed row group at entry {pos}")
        pos += 1 + c
        nblocks += c
    if pos != idx.size:
        raise ValueError(f"array '{cname}' has {idx.size} entries, its row groups account for {pos}")
    w["gru_a_idx"] = idx
    w["gru_a_w"] = take("gru_a_w", (nblocks, 4, 8))
    w["gru_b_bias"] = take("gru_b_bias", (2, 3 * nb))
    w["gru_b_w_in"] = take("gru_b_w_in", (na, 3 * nb))
    w["gru_b_w_rec"] = take("gru_b_w_rec", (nb, 3 * nb))
    w["dual_fc_bias"] = take("dual_fc_bias", (2 * d.dual_fc_out,))
    w["dual_fc_w"] = take("dual_fc_w", (d.dual_fc_out, 2, nb))
    w["dual_fc_factor"] = take("dual_fc_factor", (2 * d.dual_fc_out,))
    return w


def convert(path_in: str, path_out: str, names: Optional[Dict[str, str]] = None, gru_a_order: int = 0) -> int:
    """gru_a_order: which association order of compute_sparse_gru's z/r pre-activation the xiph revision the weights
    came with uses (include/dss_lpcnet_blob.h): 0 = input before the blocks (nnet.c 2021), 1 = blocks first (2019-20).
    Writes the blob and, when the source has ``#ifdef DOT_PROD`` branches, ``<path_out>.dotprod.npz`` with that side's
    arrays (int8 weights, scales, subias): kept for the day a xiph vector says the reference's build uses them."""
    with open(path_in, "r", errors="replace") as f:
        text = f.read()
    w = weights_from_nnet_data(text, names=names)
    _, dp, branches = parse_c_arrays_both(text)
    blob = pack_blob(w, gru_a_order=gru_a_order, source_branches=branches)
    with open(path_out, "wb") as f:
        f.write(blob)
    if dp:
        np.savez(path_out + ".dotprod.npz", **dp)
    return len(blob)


def kernel_fit(blob: bytes) -> dict:
    """Which sample-rate kernel the model will run on and how much LDS its image takes, from the host-only layout check
    of the library (``dss_selftest_fast_layout``: no GPU needed)."""
    import ctypes
    from . import _lib
    L = _lib.load()
    info = (ctypes.c_int * 8)()
    _lib.check(L.dss_selftest_fast_layout(blob, len(blob), info))
    keys = ("fast_path", "zr_blocks_max", "h_blocks_max", "lds_bytes", "zr_register_slots", "tail_blocks", "mismatches", "oob")
    out = dict(zip(keys, list(info)))
    out["kernel"] = {0: "generic (GRU A blocks streamed from L2: 3-4x slower)", 1: "CU-resident",
                     2: "CU-resident with tail paths (some rows exceed the register slots: 1.1-1.6x slower)"}[out["fast_path"]]
    return out


XIPH_CHECKS = """Before trusting a blob converted from a xiph/LPCNet tree, run these three checks IN THAT TREE (tests/golden/README.md has
the whole recipe; DESIGN.md section 2 says why they decide whether this build answers the right program):
  1. gcc -O2 -dM -E -Iinclude -Isrc src/nnet.c | grep -E "DOT_PROD|__SSE2__|__AVX__"
        DOT_PROD defined => the reference's build runs the int8 arithmetic; the blob's float weights are NOT what it runs
  2. grep -n rcp_ps src/vec_avx.h
        a hit => that build's sigmoid / tanh use the approximate reciprocal: no unique output, no +-1 LSB claim possible
  3. grep -n '#include "vec' src/nnet.c src/vec.h
        which vec*.h the build really includes (generic vec.h = what the oracle and the kernels restate)"""


def inspect_xiph_tree(src_dir: str) -> dict:
    """What can be read off the tree nnet_data.c lies in, without a compiler: does its vec.h hand a plain x86-64 build (which
    defines __SSE2__ but not __AVX__) to vec_avx.h, and does that header use the approximate reciprocal?
    {"vec_h": found?, "selects_avx_on_default_x86_64": True / False / None (unknown), "condition": text, "rcp_ps": True / False / None}"""
    import os
    out = {"vec_h": False, "selects_avx_on_default_x86_64": None, "condition": None, "rcp_ps": None}
    vec = os.path.join(src_dir, "vec.h")
    if os.path.exists(vec):
        out["vec_h"] = True
        with open(vec, "r", errors="replace") as f:
            lines = f.read().splitlines()
        cond = None
        for i, ln in enumerate(lines):
            if re.search(r'#\s*include\s*"vec_avx\.h"', ln):
                for j in range(i - 1, -1, -1):                     # the #if / #elif that guards it
                    if re.match(r"\s*#\s*(if|elif|ifdef)\b", lines[j]):
                        cond = lines[j].strip()
                        break
                break
        out["condition"] = cond
        if cond is None:
            out["selects_avx_on_default_x86_64"] = False           # vec.h never includes vec_avx.h
        else:
            out["selects_avx_on_default_x86_64"] = bool(re.search(r"__SSE2?__|__SSE4|__x86_64__|__SSSE3__", cond)) and True or \
                (False if re.search(r"__AVX", cond) else None)
    avx = os.path.join(src_dir, "vec_avx.h")
    if os.path.exists(avx):
        with open(avx, "r", errors="replace") as f:
            out["rcp_ps"] = "rcp_ps" in f.read()
    return out


def main(argv=None) -> int:
    import os
    argv = list(sys.argv[1:] if argv is None else argv)
    i_know = "--i-know" in argv
    argv = [a for a in argv if a != "--i-know"]
    if len(argv) not in (2, 3):
        print("usage: python -m dss_amd.nnet_data <nnet_data.c> <out.blob> [gru_a_order: 0 (default, xiph 2021) | 1 (2019-20)] [--i-know]",
              file=sys.stderr)
        return 2
    print(XIPH_CHECKS)
    tree = inspect_xiph_tree(os.path.dirname(os.path.abspath(argv[0])))
    if not tree["vec_h"]:
        print("\nNOTE: no vec.h beside " + argv[0] + ": the checks above could not even be pre-read here.  Run them in the tree the file "
              "came from before any parity claim.")
    else:
        print(f"\nvec.h beside the source: vec_avx.h is included under `{tree['condition']}`" if tree["condition"] else
              "\nvec.h beside the source never includes vec_avx.h (generic path)")
        if tree["rcp_ps"]:
            print("vec_avx.h uses _mm*_rcp_ps (approximate reciprocal): a build that takes it has no unique output")
        if tree["selects_avx_on_default_x86_64"] and not i_know:
            print("\nREFUSED: this tree hands a plain x86-64 build (__SSE2__ is always defined there) to vec_avx.h, i.e. to DOT_PROD / int8 "
                  "arithmetic -- not what the blob, the oracle and the kernels implement.  Run check 1 to be sure; if the reference's build "
                  "really compiles the generic path (or you want the float blob anyway), repeat with --i-know.", file=sys.stderr)
            return 3
        if tree["selects_avx_on_default_x86_64"] is None:
            print("could not tell from vec.h's condition whether a plain x86-64 build takes vec_avx.h: run check 1")
    n = convert(argv[0], argv[1], gru_a_order=int(argv[2]) if len(argv) == 3 else 0)
    print(f"wrote {argv[1]}: {n} bytes")
    from .lpcnet_weights import blob_source_branches
    with open(argv[1], "rb") as f:
        br = blob_source_branches(f.read(96))
    print("source branches: " + {1: "float only (no #ifdef DOT_PROD pairs)", 3: "float AND DOT_PROD (int8) -- the blob holds the float "
          "branch, the DOT_PROD arrays are in " + argv[1] + ".dotprod.npz; check which one the reference's build compiles "
          "(check 1 above) before trusting parity"}.get(br, str(br)))
    try:
        with open(argv[1], "rb") as f:
            fit = kernel_fit(f.read())
        print(f"sample-rate kernel: {fit['kernel']}; z/r blocks per row group <= {fit['zr_blocks_max']}, h blocks <= "
              f"{fit['h_blocks_max']}, LDS image {fit['lds_bytes']} B")
    except Exception as e:           # the library may not be built on the machine that converts
        print(f"(kernel fit not checked: {e})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
