"""Development aid: uniform calls of several sizes on the automatic kernel choice, on the one-utterance kernel and on the pair
kernel (sample-kernel time from the library's HIP events); the three must agree bit for bit."""
import os, sys, time
sys.path.insert(0, "delayed-speech-synthesis_amd")
import numpy as np, torch
from dss_amd.lpcnet import LPCNetBatch
from dss_amd.lpcnet_weights import synthetic_features
F = 100
for B in (256, 520, 600, 768, 800, 1024, 1100):
    feats = torch.from_numpy(np.stack([synthetic_features(b % 16, F) for b in range(B)])).cuda()
    out = torch.empty((B, F * 160), dtype=torch.int16, device="cuda")
    res = []
    for mode in (0, 1, 2):
        gpu = LPCNetBatch(B, F); gpu.set_multi(mode)
        gpu.synthesize_torch(feats, out=out); torch.cuda.synchronize()
        gpu.reset(); gpu.enable_timing(True)
        gpu.synthesize_torch(feats, out=out); torch.cuda.synchronize()
        res.append((gpu.kernel_ms(0), int(out.to(torch.int64).sum())))
        del gpu
    assert res[0][1] == res[1][1] == res[2][1]
    print(f"B={B}: auto {res[0][0]:.1f} ms, one per workgroup {res[1][0]:.1f} ms, two per workgroup {res[2][0]:.1f} ms", flush=True)
