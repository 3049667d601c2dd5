"""Development aid: per-phase cycle shares of the sample kernel (diagnostic build with s_memtime stamps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np, torch
from dss_amd import _lib
from dss_amd.lpcnet import LPCNetBatch
from dss_amd.lpcnet_weights import synthetic_features
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = 20
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # utterances per workgroup: 1 (latency kernel), 2 (pair kernel)
feats = np.stack([synthetic_features(b, F) for b in range(B)])
gpu = LPCNetBatch(B, F)
gpu.set_multi(MODE)
gpu.enable_trace(2)
gpu.synthesize(feats)
n = (F - 2) * 160
raw = np.empty((F * 160,), np.float32)
names = ["P1 work", "wait A", "A->B (GRU A)", "B->C (GRU B)", "C->D (FC)", "P6 / idle"]
_lib.check(gpu._L.dss_lpcnet_batch_tap(gpu._h, 0, 4, raw.ctypes.data, raw.size))   # trace_pcm holds the stamps
if MODE == 2:
    B = (B + 1) // 2                                       # one record per workgroup
st = raw[: B * 6].reshape(B, 6) / n
print("wave 7 (scalar role) view, cycles per sample, mean over utterances:")
print("  " + "  ".join(f"{names[k]}={st[:, k].mean():8.1f}" for k in range(6)), " total", st.sum(axis=1).mean())
print("  min/max total over utterances:", st.sum(axis=1).min(), st.sum(axis=1).max())
if MODE == 2 and B == 1:                                   # one workgroup: the relay waves' own stamps inside B..C
    print(f"  GRU B relay, from barrier B: wave 6 hands over for the last time at {raw[65] / n:.1f} and reaches C at {raw[66] / n:.1f}, "
          f"wave 7 ends the chain at {raw[64] / n:.1f} and reaches C at {raw[67] / n:.1f}")
if MODE == 1 and B == 1:      # one workgroup: the relay waves' own way points inside B..C
    w7 = raw[64:70] / n
    w6 = raw[72:76] / n
    print(f"  GRU B relay, cycles from barrier B: wave 6 segment 1 summed + published {w6[0]:.0f} | wave 7 segment 2 products ready {w7[0]:.0f}, "
          f"has the sums {w7[1]:.0f}, segment 2 summed + published {w7[2]:.0f} | wave 6 segment 3 products ready {w6[1]:.0f}, has the sums {w6[2]:.0f}, "
          f"summed + published {w6[3]:.0f} | wave 7 segment 4 products ready {w7[3]:.0f}, has the sums {w7[4]:.0f}, summed {w7[5]:.0f}")
rawA = np.empty((F * 160,), np.float32)
chunks = []
for k in range((B * 48 + rawA.size - 1) // rawA.size):
    _lib.check(gpu._L.dss_lpcnet_batch_tap(gpu._h, k, 3, rawA.ctypes.data, rawA.size))
    chunks.append(rawA.copy())
sa = np.concatenate(chunks)[: B * 48].reshape(B, 6, 8) / n
namesA = ["A->products", "emb wait+gz", "z/r sums", "activations", "wait B", "h chain", "wait C", "FC+wait D"]
print("role A view (cycles per sample, mean over utterances), per wave:")
for w in range(6):
    print(f"  wave {w}: " + "  ".join(f"{namesA[k]}={sa[:, w, k].mean():7.1f}" for k in range(8)))
