"""Development aid: the two one-utterance-per-workgroup sample kernels (lpcnet_sample.hip / lpcnet_sample_pkh.hip) timed
alternately in one process on one box, with a bit-for-bit comparison of their PCM.

    python tools/lat_ab.py [batch=256] [frames=100] [rounds=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
import torch
from dss_amd import _lib
from dss_amd.lpcnet import LPCNetBatch
from dss_amd.lpcnet_weights import synthetic_features

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = _lib.load()
feats = torch.from_numpy(np.stack([synthetic_features(b, F) for b in range(B)])).cuda()
gpu = LPCNetBatch(B, F)
gpu.set_multi(1)
out = {1: torch.empty((B, F * 160), dtype=torch.int16, device="cuda"), 2: torch.empty((B, F * 160), dtype=torch.int16, device="cuda")}
ms = {1: [], 2: []}
for it in range(R):
    for which in (1, 2):
        _lib.check(L.dss_selftest_lpcnet_latency_kernel(which))
        gpu.reset()
        gpu.enable_timing(True)
        gpu.synthesize_torch(feats, out=out[which])
        torch.cuda.synchronize()
        ms[which].append(gpu.kernel_ms(0))
_lib.check(L.dss_selftest_lpcnet_latency_kernel(0))
same = bool(torch.equal(out[1], out[2]))
print(f"B={B} F={F}: lpcnet_sample.hip {['%.2f' % m for m in ms[1]]} ms | lpcnet_sample_pkh.hip {['%.2f' % m for m in ms[2]]} ms | "
      f"ratio {np.median(ms[2]) / np.median(ms[1]):.4f} | PCM identical: {same}")
if not same:
    d = (out[1] != out[2]).nonzero()
    print("first mismatches (utt, sample):", d[:8].tolist(), "rows differing:", int((out[1] != out[2]).any(dim=1).sum()))
    sys.exit(1)
