"""Development aid: sample-kernel time per sample for short calls (the streaming tick: 128 rows x 4 frames, state carried) against long ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np, torch
from dss_amd import _lib
from dss_amd.lpcnet import LPCNetBatch, load_model
from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features
load_model(synthetic_blob(0))
L = _lib.load()
for B, F, calls in ((128, 100, 3), (128, 4, 60), (128, 1, 60), (256, 100, 3), (1, 1, 60)):
    gpu = LPCNetBatch(B, F)
    feats = torch.from_numpy(np.stack([synthetic_features(b, 104) for b in range(B)])).cuda()
    L.dss_lpcnet_batch_enable_timing(gpu._h, 1)
    ms = []
    gpu.synthesize_torch(feats[:, :F].contiguous())            # first call: silent frames
    for c in range(calls):
        gpu.synthesize_torch(feats[:, 4:4 + F].contiguous())
        torch.cuda.synchronize()
        ms.append(L.dss_lpcnet_batch_kernel_ms(gpu._h, 0))
    m = float(np.median(ms))
    print(f"{B} rows x {F} frames per call: sample kernel {m * 1e3:.1f} us = {m * 1e3 / (F * 160):.3f} us per sample")
