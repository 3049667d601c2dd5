"""Development aid: cycles per slot segment of the throughput kernel (diagnostic build with s_memtime stamps), workgroup 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
from dss_amd import _lib, lpcnet
from dss_amd.lpcnet_weights import synthetic_features
U = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
F = 12
lpcnet.load_model(synthetic=True)
feats = np.stack([synthetic_features(b % 64, F) for b in range(B)])
gpu = lpcnet.LPCNetBatch(B, F)
gpu.set_multi(U)
gpu.enable_trace(2)
gpu.synthesize(feats)
raw = np.empty((F * 160,), np.float32)
_lib.check(gpu._L.dss_lpcnet_batch_tap(gpu._h, 0, 4, raw.ctypes.data, raw.size))
slots = U * ((F - 2) * 160 + 1) + 2
st = raw[:64].reshape(8, 8) / slots
namesA = ["FC | spec(4,5)", "wait bits", "walk+emb issue", "h chain", "products", "wait reads+sums+gates", "barrier wait"]
for w in range(6):
    print(f"wave {w}: " + "  ".join(f"{namesA[q]}={st[w, q]:7.1f}" for q in range(7)), f" total {st[w, :7].sum():.0f}")
print("wave 6: " + "  ".join(f"{n}={st[6, q]:7.1f}" for q, n in enumerate(["chain half 1", "speculation", "barrier wait"])), f" total {st[6, :3].sum():.0f}")
print("wave 7: " + "  ".join(f"{n}={st[7, q]:7.1f}" for q, n in enumerate(["chain half 2+gates", "bookkeeping", "barrier wait"])), f" total {st[7, :3].sum():.0f}")
