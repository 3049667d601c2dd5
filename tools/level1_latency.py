#!/usr/bin/env python3
"""Latency of the Level-1 drop-in path: what an UNCHANGED decode_online.py pays per call.

  * LPCNet.LPCNet().synthesize(features[20]) -> int16[160]: p50 / p99 over 1000 calls
  * a 150-frame segment through the reference's loop (local/units.py:534-535): [synthesize(row) for row in segment]
  * the same segment through dss_amd.units.DelayedLPCNetVocoder (Level 2: one launch per segment)
"""
import asyncio
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np

os.environ.setdefault("DSS_LPCNET_SYNTHETIC", "1")
import LPCNet
from dss_amd import units as U
from dss_amd.lpcnet_weights import synthetic_features

net = LPCNet.LPCNet()
f = synthetic_features(1, 1200)
for t in range(50):
    net.synthesize(f[t])
lat = []
for t in range(50, 1050):
    t0 = time.perf_counter()
    net.synthesize(f[t])
    lat.append((time.perf_counter() - t0) * 1e3)
seg = f[:150]
t0 = time.perf_counter()
pcm = np.hstack([net.synthesize(row) for row in seg])
loop_ms = (time.perf_counter() - t0) * 1e3

voc = U.DelayedLPCNetVocoder()
voc.initialize()


async def drive(gen):
    return [m async for m in gen]

asyncio.run(drive(voc.synthesize(U.ClosedLoopMessage(data=seg, fs=100))))
t0 = time.perf_counter()
asyncio.run(drive(voc.synthesize(U.ClosedLoopMessage(data=seg, fs=100))))
unit_ms = (time.perf_counter() - t0) * 1e3
print(json.dumps({"per_call_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                                  "mean": float(np.mean(lat)), "calls": len(lat)},
                  "segment_150_frames_ms": {"reference_loop_level1": loop_ms, "vocoder_unit_level2": unit_ms,
                                            "audio_ms": 1500.0}}))
