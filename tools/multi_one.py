#!/usr/bin/env python3
"""One configuration of the sample-rate kernel for profiling:  python tools/multi_one.py <mode: -1|3|4|0> [batch] [frames] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
import torch
from dss_amd import lpcnet
from dss_amd.lpcnet_weights import synthetic_features
mode = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
F = int(sys.argv[3]) if len(sys.argv) > 3 else 100; steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lpcnet.load_model(synthetic=True)
feats = torch.from_numpy(np.stack([synthetic_features(b % 64, F) for b in range(B)])).cuda()
out = torch.empty((B, F * 160), dtype=torch.int16, device="cuda")
dec = lpcnet.LPCNetBatch(B, F); dec.set_multi(mode)
dec.reset_async(); dec.synthesize_torch(feats, out=out); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    dec.reset_async(); dec.synthesize_torch(feats, out=out)
torch.cuda.synchronize()
print("mode", mode, "batch", B, "ms/step", (time.perf_counter() - t0) / steps * 1e3)
