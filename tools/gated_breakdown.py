"""Development aid: where a gated streaming tick goes (row f4; 128 streams, 40-sample packets, no segment closing).

    python tools/gated_breakdown.py [streams=128] [ticks=300]

Prints p50 / p99 of the whole tick (host packet in -> event counts on the host) with the VAD on the library's kernel and on
the PyTorch-ROCm module, and the device time of the three kernels of the non-closing tick (HIP events on torch's stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np
import torch
from dss_amd import lpcnet
from dss_amd.lpcnet_weights import synthetic_blob
from dss_amd.pipeline import GatedStreamingPipeline

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lpcnet.load_model(synthetic_blob(0))
rng = np.random.default_rng(0)
pk = [rng.standard_normal((S, 40, 64)) * 2.0 for _ in range(16)]      # quiet input: the seeded detector rarely closes a segment


class _NeverSpeech(torch.nn.Module):          # a detector that never says speech: the tick without a closing segment
    def create_new_initial_state(self, batch_size, device="cpu", req_grad=False):
        return (torch.zeros(1, device=device), torch.zeros(1, device=device))

    def forward(self, x, state=None):
        out = torch.zeros(x.shape[0], x.shape[1], 2, device=x.device)
        out[..., 0] = 1.0
        return out, state


def run(tag, **kw):
    p = GatedStreamingPipeline(S, 64, **kw)
    lat, closed = [], 0
    for k in range(T + 20):
        t0 = time.perf_counter()
        out = p.push(pk[k % len(pk)])
        torch.cuda.synchronize()
        if k >= 20:
            lat.append((time.perf_counter() - t0) * 1e3)
            closed += len(out)
    lat = np.asarray(lat)
    print(f"{tag}: tick p50 {np.percentile(lat, 50):.3f} ms  p99 {np.percentile(lat, 99):.3f} ms  ({closed} segments closed in {T} ticks)", flush=True)
    return p


p = run("VAD on the library's kernel (csrc/vad_lstm.hip)  ")
run("VAD as the PyTorch-ROCm module (MIOpen LSTM)     ", use_vad_kernel=False)
run("no VAD work at all (a module that says 'silence')", vad=_NeverSpeech())
# device time of the pieces of a non-closing tick
z = torch.from_numpy(pk[0]).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
acc = np.zeros(3)
for k in range(50):
    p._in.copy_(z)
    ev[0].record()
    hga = p.hga.extract_torch(p._in, apply_log=True)
    ev[1].record()
    labels = p.vad_gpu.step_torch(hga)
    ev[2].record()
    p.gate.push_torch(hga, labels)
    ev[3].record()
    torch.cuda.synchronize()
    if k >= 10:
        acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
acc /= 40
print(f"device time per tick: HGA {acc[0] * 1e3:.1f} us | VAD kernel {acc[1] * 1e3:.1f} us | gate kernel + event read-back {acc[2] * 1e3:.1f} us")
