"""Timing of the ragged batch call and of the gated streaming tick (development aid; bench.py is the contract).

  * 1024 utterances of 0.5-3 s (uniform) with a fresh decoder each: one ragged launch (longest first) against the
    padded uniform launch that synthesises every row to the longest length.
  * GatedStreamingPipeline, 128 streams: time of a tick on which no segment closes (HGA + VAD LSTM + gate + the
    event read-back), which is what every 40 ms packet costs between speech segments.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
import torch

from dss_amd import lpcnet
from dss_amd.lpcnet_weights import synthetic_features
from dss_amd.pipeline import GatedStreamingPipeline

lpcnet.load_model(synthetic=True)
rng = np.random.default_rng(0)
n = 1024
counts = rng.integers(50, 301, n)
fmax = int(counts.max())
base = synthetic_features(0, fmax)
feats = torch.from_numpy(np.stack([base] * n)).cuda()
order = np.argsort(-counts, kind="stable")
dec = lpcnet.LPCNetBatch(n, fmax)
out = torch.empty((n, fmax * 160), dtype=torch.int16, device="cuda")
for name, c, mode in (("ragged, longest first", counts[order], 0), ("ragged, longest first, one utterance per workgroup", counts[order], 1),
                      ("ragged, arrival order", counts, 0), ("padded to the longest", None, 0)):
    dec.set_multi(mode)                        # 0: the library's rule (two utterances per workgroup beyond one row per CU)
    for it in range(2):
        dec.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        if c is None:
            dec.synthesize_torch(feats, out=out)
        else:
            dec.synthesize_ragged_torch(feats, c, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    audio = counts.sum() * 0.01
    print(f"{name:52s}: {dt * 1e3:8.1f} ms for {audio:.0f} s of audio in {n} utterances -> {audio / dt:7.0f} x RT", flush=True)
del dec, feats, out

S = 128
pipe = GatedStreamingPipeline(S, 64)
lat = []
for k in range(120):
    pk = rng.standard_normal((S, 40, 64)) * 50.0
    t = time.perf_counter()
    segs = pipe.push(pk)
    lat.append((time.perf_counter() - t) * 1e3)
    assert not segs or k > 20
lat = np.asarray(lat[20:])
print(f"gated tick, {S} streams, no segment closing: p50 {np.percentile(lat, 50):.2f} ms, p99 {np.percentile(lat, 99):.2f} ms "
      f"per 40 ms packet", flush=True)
