"""Development aid: where the streaming tick's latency goes (config 5)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))
import numpy as np, torch
from dss_amd.pipeline import StreamingPipeline
from dss_amd.lpcnet import FRAME_SIZE
S = 128
sp = StreamingPipeline(S)
rng = np.random.default_rng(0)
for _ in range(5):
    sp.push(rng.standard_normal((S, 40, 64)) * 50)
acc = np.zeros(6)
N = 50
for _ in range(N):
    pk = rng.standard_normal((S, 40, 64)) * 50
    t = [time.perf_counter()]
    sp._in.copy_(torch.from_numpy(pk)); torch.cuda.synchronize(); t.append(time.perf_counter())
    hga = sp.hga.extract_torch(sp._in, apply_log=True); torch.cuda.synchronize(); t.append(time.perf_counter())
    with torch.no_grad():
        feats, _ = sp.decoder(hga.to(torch.float32), sp.decoder.create_new_initial_state(batch_size=S, device="cuda"))
    torch.cuda.synchronize(); t.append(time.perf_counter())
    pcm = sp.vocoder.synthesize_torch(feats.contiguous()); torch.cuda.synchronize(); t.append(time.perf_counter())
    r = pcm.cpu(); t.append(time.perf_counter())
    r = r.numpy(); t.append(time.perf_counter())
    acc += np.diff(t)
print("ms per tick: h2d %.3f | hga %.3f | bilstm %.3f | lpcnet %.3f | d2h %.3f | copy %.3f | total %.3f" % (*(acc / N * 1e3), acc.sum() / N * 1e3))
sp.vocoder.enable_timing(True)
for _ in range(10):
    sp.push(rng.standard_normal((S, 40, 64)) * 50)
print("lpcnet kernels: sample %.3f ms, frame %.3f ms" % (sp.vocoder.kernel_ms(0), sp.vocoder.kernel_ms(1)))
