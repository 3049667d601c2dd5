"""Development aid: the decoder kernels (csrc/bilstm_decoder.hip) against the PyTorch-ROCm module (MIOpen LSTM), device time per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np, torch
from dss_amd.decoder import BiLstmDecoderGPU
from dss_amd.models import BidirectionalSpeechSynthesisModel

torch.manual_seed(0)
m = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=64).eval().cuda()


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for S, T in ((128, 4), (64, 104), (1, 346), (1024, 4)):
    k = BiLstmDecoderGPU(S, T, m)
    z = torch.randn((S, T, 64), dtype=torch.float64, device="cuda")
    with torch.no_grad():
        tm = timed(lambda: m(z.to(torch.float32), m.create_new_initial_state(batch_size=S, device="cuda")))
    tk = timed(lambda: k(z))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        k(z)
    tg = timed(g.replay)
    print(f"{S} streams x {T} frames: kernels {tk:.3f} ms (graph replay {tg:.3f}), PyTorch-ROCm module {tm:.3f} ms")
