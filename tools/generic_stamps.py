"""Development aid: per-phase cycles of the GENERIC sample-rate kernel (diagnostic build with s_memtime stamps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
from dss_amd import _lib, lpcnet
from dss_amd.lpcnet_weights import synthetic_features
B, F = 256, 12
lpcnet.load_model(synthetic=True)
feats = np.stack([synthetic_features(b, F) for b in range(B)])
gpu = lpcnet.LPCNetBatch(B, F)
gpu.enable_trace(18)          # 16: generic kernel, 2: phase stamps
gpu.synthesize(feats)
n = (F - 2) * 160
raw = np.empty((F * 160,), np.float32)
_lib.check(gpu._L.dss_lpcnet_batch_tap(gpu._h, 0, 4, raw.ctypes.data, raw.size))
st = raw[:48].reshape(8, 6) / n
names = ["P1 (wave 7 scalar)", "wait A", "A->B (GRU A)", "B->C (GRU B)", "C->D (FC)", "P6 (walk, pcm)"]
for w in (0, 5, 6, 7):
    print(f"wave {w}: " + "  ".join(f"{names[k]}={st[w, k]:8.1f}" for k in range(6)), f" total {st[w].sum():.0f}")
