"""Workload for rocprofv3 over the HGA kernels: 1024 streams x 1.04 s x 64 ch, the plain call and the raw-packet call
(front end + z-score), each 5 times, through the form DSS_HGA_PATH selects (0 default: fused kernel + separate front end, 2: three launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np, torch
from dss_amd.hga import HgaExtractorGPU
from dss_amd.electrodes import reference_frontend
from dss_amd.synthetic import synthetic_ecog
PATH = int(os.environ.get("DSS_HGA_PATH", "0"))
S = 1024
x = torch.from_numpy(np.stack([synthetic_ecog(1000 + b % 8, 1040, 64) for b in range(S)])).cuda()
ex = HgaExtractorGPU(S, 64)
ex._force_path(PATH)
for _ in range(5):
    ex.reset(); ex.extract_torch(x, apply_log=True)
raw = torch.from_numpy(np.random.default_rng(1).standard_normal((S, 1040, 129)) * 50).cuda()
ex = HgaExtractorGPU(S, 64)
ex._force_path(PATH)
ex.set_frontend(129, *reference_frontend())
if True:
    ex.set_zscore(np.zeros(64), np.ones(64))
for _ in range(5):
    ex.reset(); ex.extract_raw_torch(raw)
torch.cuda.synchronize()
