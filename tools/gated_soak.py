"""Development aid (GPU box): a longer paced run of the gated streaming mode -- N passes over tools/gated_leg.py's 10.4 s of input at
the 40 ms cadence -- checking that every closed segment comes back exactly once and in order per stream, that the queue's event and
pool recycling holds up, and that host memory does not grow."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("DSS_LPCNET_SYNTHETIC", "1")
import numpy as np
import psutil

import gated_leg
from dss_amd.pipeline import GatedStreamingPipeline

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
packets = gated_leg.make_input()
gp = GatedStreamingPipeline(gated_leg.S, 64, channel_means=np.full(64, 5.0), vad=gated_leg.detector(), max_segment_frames=2000)
proc = psutil.Process()
last_prev = {}
n_seg, ticks, t0 = 0, 0, time.perf_counter()
tick_ms = []


def take(got):
    global n_seg
    for s, prev, pcm in got:
        assert prev > last_prev.get(s, -10**9), (s, prev, last_prev.get(s))   # a stream's segments in closing order (the first may start "before" frame 0: ring context)
        last_prev[s] = prev
        assert len(pcm) % 160 == 0 and len(pcm) > 0
        n_seg += 1


for p in range(passes):
    rss0 = proc.memory_info().rss
    for k in range(len(packets)):
        due = t0 + 0.04 * ticks
        while time.perf_counter() < due:
            take(gp.poll())
            time.sleep(0.0005)
        t1 = time.perf_counter()
        take(gp.push(packets[k]))
        tick_ms.append((time.perf_counter() - t1) * 1e3)
        ticks += 1
    print(f"pass {p}: {ticks} ticks, {n_seg} segments back of {gp.segments_closed} closed, in flight {gp.queue.in_flight}, "
          f"events created {len(gp.queue._all_events)}, free pool rows {len(gp.queue._free_rows)}, "
          f"tick p50 {np.percentile(tick_ms, 50):.3f} p99 {np.percentile(tick_ms, 99):.3f} max {max(tick_ms):.3f} ms, "
          f"rss {proc.memory_info().rss / 2**20:.0f} MiB ({(proc.memory_info().rss - rss0) / 2**20:+.1f})", flush=True)
take(gp.flush())
assert n_seg == gp.segments_closed and gp.queue.in_flight == 0, (n_seg, gp.segments_closed)
lat = gp.queue.latencies_ms
print(f"done: {n_seg} segments, close -> PCM p50 {np.percentile(lat, 50):.1f} p99 {np.percentile(lat, 99):.1f} max {max(lat):.1f} ms, wall {time.perf_counter() - t0:.1f} s "
      f"for {ticks * 0.04:.1f} s of streams")
gp.close()
