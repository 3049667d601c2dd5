"""Which sample kernel a model gets, how much of the CU it uses (dss_lpcnet_model_info) and what that costs: the seeded
model and models with increasingly skewed sparsity, 256 x 1 s, sample-kernel time from the library's HIP events, next
to the generic kernel on the same model.  Run on the GPU box:  python tools/model_fit.py"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
from dss_amd import lpcnet
from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features

warnings.simplefilter("ignore")
B, F = 256, 100
feats = np.stack([synthetic_features(b, F) for b in range(B)])
for seed, skew in ((0, 0.0), (0, 0.02), (7, 0.05), (0, 0.1), (7, 0.3)):
    lpcnet.load_model(blob=synthetic_blob(seed, skew=skew))
    ms = {}
    for name, trace in (("chosen", 0), ("generic", 16)):
        dec = lpcnet.LPCNetBatch(B, F)
        if trace:
            dec.enable_trace(trace)
        dec.enable_timing(True)
        t = []
        for _ in range(3):
            dec.reset()
            dec.synthesize(feats)
            t.append(dec.kernel_ms(0))
        ms[name] = min(t)
        dec.close()
    info = lpcnet.model_info()
    print(f"seed {seed} skew {skew}: fast_path {info['fast_path']} zr_max {info['zr_slots_max']} h_max {info['h_slots_max']} "
          f"lds {info['h_lds_bytes']} B | {info['kernel']} {ms['chosen']:.1f} ms, generic {ms['generic']:.1f} ms", flush=True)
