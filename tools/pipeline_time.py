"""Development aid: timing of the HGA kernel alone and of the config-3 segment pipeline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from dss_amd.hga import HgaExtractorGPU, design_filters
from dss_amd.pipeline import SegmentPipeline
from dss_amd.synthetic import synthetic_ecog

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

for S in (64, 1024):
    x = torch.from_numpy(np.stack([synthetic_ecog(1000 + b % 8, 1040, 64) for b in range(S)])).cuda()
    ex = HgaExtractorGPU(S, 64)
    def run():
        ex.reset(); ex.extract_torch(x, apply_log=True)
    dt = timeit(run)
    gb = S * (1040 * 64 * 8 + 100 * 64 * 8) / 1e9
    print(f"HGA {S} x 1.04 s x 64 ch: {dt*1e3:.3f} ms -> {S*1.04/dt:.0f} stream-seconds/s, {gb/dt:.1f} GB/s of {gb*1e3:.1f} MB algorithmic")
B = 64
ecog = torch.from_numpy(np.stack([synthetic_ecog(1000 + b, 1040, 64) for b in range(B)])).cuda()
pipe = SegmentPipeline(B)
dt = timeit(lambda: pipe(ecog), 5)
print(f"config 3 pipeline, {B} segments: {dt*1e3:.2f} ms per batch -> {B*16000/dt/16000:.0f} x RT")
import oracle_api
orc = oracle_api.Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
hg, fh, zi_hg, zi_fh = design_filters(1000)
xe = synthetic_ecog(1000, 1040, 64)
t = time.perf_counter()
for _ in range(20):
    orc.extractor({"sos_hg": hg, "sos_fh": fh, "zi_hg": zi_hg, "zi_fh": zi_fh}, 64).extract(xe)
dt = (time.perf_counter() - t) / 20
print(f"CPU oracle HGA, 1 core: {dt*1e3:.2f} ms per 1.04 s x 64 ch trial -> {1.04/dt:.0f} stream-seconds/s")
