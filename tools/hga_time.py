"""Development aid: HGA extractor timing: hga_fused_kernel (default), the three-launch form (DSS_HGA_PATH=2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np, torch
from dss_amd.hga import HgaExtractorGPU
from dss_amd.electrodes import reference_frontend
from dss_amd.synthetic import synthetic_ecog

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

PATH = int(os.environ.get("DSS_HGA_PATH", "0"))          # 0 default (fused), 2 three launches
tag = {0: "fused", 1: "fused", 2: "3 launches"}[PATH]
for S in (64, 1024):
    x = torch.from_numpy(np.stack([synthetic_ecog(1000 + b % 8, 1040, 64) for b in range(S)])).cuda()
    ex = HgaExtractorGPU(S, 64)
    ex._force_path(PATH)
    def run():
        ex.reset(); ex.extract_torch(x, apply_log=True)
    dt = timeit(run)
    gb = S * (1040 * 64 * 8 + 100 * 64 * 8) / 1e9
    gf = S * 1040 * 64 * 16 * 9 / 1e9
    print(f"[{tag}] HGA {S} x 1.04 s x 64 ch: {dt*1e3:.3f} ms -> {S*1.04/dt:.0f} stream-s/s, {gb/dt:.0f} GB/s algorithmic, {gf/dt/1e3:.2f} TFLOP/s fp64 (no-FMA peak 39.3)")
S = 128
ex = HgaExtractorGPU(S, 64)
ex._force_path(PATH)
pk = torch.from_numpy(np.random.default_rng(0).standard_normal((S, 40, 64)) * 50).cuda()
ex.extract_torch(pk)
print(f"[{tag}] streaming tick, 128 streams x 40-sample packet: {timeit(lambda: ex.extract_torch(pk), 50)*1e3:.4f} ms")
raw = torch.from_numpy(np.random.default_rng(1).standard_normal((1024, 1040, 129)) * 50).cuda()
ex = HgaExtractorGPU(1024, 64)
ex._force_path(PATH)
ex.set_frontend(129, *reference_frontend())
def run2():
    ex.reset(); ex.extract_raw_torch(raw)
print(f"[{tag}] raw 129-column packets with the fused front end, 1024 x 1.04 s: {timeit(run2, 5)*1e3:.3f} ms")
