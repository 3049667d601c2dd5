"""Development probe (GPU box): does work on torch's current stream run WHILE a long vocoder launch occupies a side stream?
Times small operations on the tick's stream with and without a ~100 ms ragged vocoder call in flight on a lane."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
os.environ.setdefault("DSS_LPCNET_SYNTHETIC", "1")
import numpy as np
import torch

from dss_amd import _lib, lpcnet
from dss_amd.lpcnet_weights import synthetic_features

L = _lib.require_gpu()
lpcnet.load_model(synthetic=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "lib"
parent = lpcnet.LPCNetBatch(8, 1)
lane = parent.create_lane(4, 300)
if mode == "torch":
    ts = torch.cuda.Stream()
    side = ts.cuda_stream
else:
    side = L.dss_stream_create()
print("mode", mode, "side stream", hex(side), "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
feats = torch.from_numpy(np.stack([synthetic_features(b, 300) for b in range(2)])).cuda()
x = torch.zeros(1 << 20, device="cuda")
h = np.zeros((128, 40, 64))
d = torch.empty((128, 40, 64), dtype=torch.float64, device="cuda")
done = L.dss_event_create()


def timed(fn, n=20):
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t0) * 1e3)
    return "p50 %.3f max %.3f ms" % (np.percentile(out, 50), max(out))


def small_kernel():
    x.add_(1.0)
    torch.cuda.current_stream().synchronize()


def h2d():
    d.copy_(torch.from_numpy(h))


def d2h():
    x[:1024].cpu()


for name, fn in (("kernel+sync", small_kernel), ("h2d 2.6MB sync", h2d), ("d2h 4KB", d2h)):
    torch.cuda.synchronize()
    idle = timed(fn)
    t0 = time.perf_counter()
    lane.synthesize_ragged_torch(feats, [300, 300], slots=[0, 1], stream=side)
    L.dss_event_record(done, side)
    t_issue = (time.perf_counter() - t0) * 1e3
    busy = timed(fn)
    q = L.dss_event_query(done)
    L.dss_event_synchronize(done)
    t_all = (time.perf_counter() - t0) * 1e3
    print(f"{name:16s} idle: {idle} | side stream busy: {busy} | issue {t_issue:.3f} ms, still running after the probes: {q == 0}, job {t_all:.1f} ms")
