#!/usr/bin/env python3
"""Time the sample-rate kernel on one GPU for a batch, latency kernel vs throughput kernel (3 / 4 utterances per workgroup).
    python tools/multi_time.py [batch] [frames]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
import torch

from dss_amd import lpcnet
from dss_amd.lpcnet_weights import synthetic_features

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lpcnet.load_model(synthetic=True)
feats = torch.from_numpy(np.stack([synthetic_features(b % 64, F) for b in range(B)])).cuda()
out = torch.empty((B, F * 160), dtype=torch.int16, device="cuda")
dec = lpcnet.LPCNetBatch(B, F)
res = {}
ref = None
for mode in (-1, 3, 4, 0):
    try:
        dec.set_multi(mode)
    except Exception as e:
        res[str(mode)] = str(e)
        continue
    for _ in range(2):
        dec.reset_async(); dec.synthesize_torch(feats, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        dec.reset_async(); dec.synthesize_torch(feats, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    pcm = out.cpu().numpy().copy()
    if ref is None:
        ref = pcm
    res[str(mode)] = {"ms_per_step": ms, "Msamples_per_s": B * F * 160 / ms / 1e3, "equal_to_latency_kernel": bool(np.array_equal(pcm, ref))}
print(json.dumps({"batch": B, "frames": F, "modes(-1 latency, 3, 4, 0 auto)": res}))
