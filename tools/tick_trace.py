"""Development aid: the streaming tick of config 5 (128 streams x one 40-sample packet) for a kernel trace.

    rocprofv3 --kernel-trace --stats -d gpurun_out/tick -- python3 tools/tick_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np, torch
from dss_amd.lpcnet import load_model
from dss_amd.lpcnet_weights import synthetic_blob
from dss_amd.pipeline import StreamingPipeline
load_model(synthetic_blob(0))
p = StreamingPipeline(128, 64, use_graph=os.environ.get("USE_GRAPH", "1") == "1")
lat = p.measure_latency(int(os.environ.get("TICKS", "200")))
print("p50 %.3f ms  p99 %.3f ms" % (np.percentile(lat[20:], 50), np.percentile(lat[20:], 99)))
