"""Quick device timing of the LPCNet batch path (development aid; bench.py is the contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
import torch
from dss_amd.lpcnet import LPCNetBatch, bytes_per_sample
from dss_amd.lpcnet_weights import synthetic_features

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # utterances per workgroup: 0 auto, 1, 2
feats = torch.from_numpy(np.stack([synthetic_features(b, F) for b in range(B)])).cuda()
gpu = LPCNetBatch(B, F)
gpu.set_multi(MODE)
gpu.enable_timing(True)
out = torch.empty((B, F * 160), dtype=torch.int16, device="cuda")
for it in range(3):
    gpu.reset()
    t = time.time()
    gpu.synthesize_torch(feats, out=out)
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"B={B} mode={MODE} iter {it}: wall {dt*1e3:.2f} ms  sample-kernel {gpu.kernel_ms(0):.2f} ms frame-kernels {gpu.kernel_ms(1):.3f} ms "
          f"-> {B*F*160/dt/16000:.0f} x RT, {B*F*160/dt*bytes_per_sample()/1e12:.2f} TB/s algorithmic", flush=True)
    gpu.enable_timing(True)
