"""Development probe (GPU box): host time of each stage of GatedStreamingPipeline.push(), split by whether segment jobs were
in flight on the side streams when the tick started."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
os.environ.setdefault("DSS_LPCNET_SYNTHETIC", "1")
import numpy as np
import torch

from dss_amd.models import UnidirectionalVoiceActivityDetector
from dss_amd.pipeline import GatedStreamingPipeline

S, ticks = 128, 120
rng = np.random.default_rng(1)
env = np.empty((S, ticks * 40))
for s_ in range(S):
    t_, loud = 0, bool(rng.integers(2))
    while t_ < env.shape[1]:
        n_ = int(rng.integers(1000, 4000))
        env[s_, t_:t_ + n_] = 60.0 if loud else 3.0
        loud, t_ = not loud, t_ + n_
packets = [(rng.standard_normal((S, 40, 64)) * env[:, 40 * k:40 * k + 40, None]) for k in range(ticks)]
torch.manual_seed(5)
vad = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=64)
with torch.no_grad():
    vad.classifier.weight[1] = -vad.classifier.weight[0]
    vad.classifier.bias.zero_()
gp = GatedStreamingPipeline(S, 64, channel_means=np.full(64, 5.0), vad=vad, max_segment_frames=1040)
stages = {}


def wrap(obj, name, tag):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        stages.setdefault(tag, []).append((time.perf_counter() - t0) * 1e3)
        return r
    setattr(obj, name, w)


wrap(gp._in, "copy_", "h2d")
wrap(gp.hga, "extract_torch", "hga")
wrap(gp.vad_gpu, "step_torch", "vad")
wrap(gp.gate, "push_torch", "gate+sync")
wrap(gp.queue, "submit", "submit")
wrap(gp.queue, "poll", "poll")
wrap(gp.queue, "_launch", "launch")
rows = []
paced = "--paced" in sys.argv          # the amplifier's 40 ms cadence, the host polling in between (as bench.py's paced leg)
t_start = time.perf_counter()
for k in range(ticks):
    if paced:
        while time.perf_counter() < t_start + 0.04 * k:
            gp.poll()
            time.sleep(0.0005)
    for v in stages.values():
        v.clear()
    busy = gp.queue.in_flight
    t0 = time.perf_counter()
    gp.push(packets[k])
    ms = (time.perf_counter() - t0) * 1e3
    rows.append((busy > 0, ms, {a: sum(b) for a, b in stages.items()}))
gp.flush()
print("paced (40 ms cadence)" if paced else "ticks back to back")
worst = sorted(rows[10:], key=lambda r: -r[1])[:5]
for r in worst:
    print("   slowest ticks: %.3f ms  " % r[1] + "  ".join("%s %.3f" % kv for kv in r[2].items()))
for flag in (False, True):
    sel = [r for r in rows[10:] if r[0] == flag]
    if not sel:
        continue
    print("jobs in flight at tick start:" if flag else "idle side streams:", len(sel), "ticks, push p50 %.3f max %.3f ms" % (np.percentile([r[1] for r in sel], 50), max(r[1] for r in sel)))
    for tag in ("h2d", "hga", "vad", "gate+sync", "submit", "poll", "launch"):
        vals = [r[2].get(tag, 0.0) for r in sel]
        print("   %-10s p50 %.3f  max %.3f" % (tag, np.percentile(vals, 50), max(vals)))
