"""A/B device timing of two builds of libdss_hip.so on the same box (development aid).

    python tools/ab_time.py path/to/a.so path/to/b.so ...

Each library is timed in its own child process (the exported symbols would clash otherwise) through the C ABI only:
batch 256 x 100 frames (AB_BATCH=n: another batch size), sample-kernel time from the library's own HIP events."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))


def child(path):
    import numpy as np
    import torch
    from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i = C.c_void_p, C.c_int
    L.dss_lpcnet_batch_create.restype = vp
    L.dss_lpcnet_batch_create.argtypes = [i, i]
    L.dss_lpcnet_batch_synthesize_dev.argtypes = [vp, vp, i, i, i, vp, vp]
    L.dss_lpcnet_batch_reset.argtypes = [vp, i]
    L.dss_lpcnet_batch_enable_timing.argtypes = [vp, i]
    L.dss_lpcnet_batch_kernel_ms.argtypes = [vp, i]
    L.dss_lpcnet_batch_kernel_ms.restype = C.c_double
    L.dss_lpcnet_load_model.argtypes = [C.c_char_p, C.c_size_t]
    blob = synthetic_blob(0)
    assert L.dss_lpcnet_load_model(blob, len(blob)) == 0
    if os.environ.get("AB_LATENCY_KERNEL") and hasattr(L, "dss_selftest_lpcnet_latency_kernel"):     # builds of the round-4 experiment
        assert L.dss_selftest_lpcnet_latency_kernel(int(os.environ["AB_LATENCY_KERNEL"])) == 0      #   (tools/experiments/) carry two latency kernels
    B, F = int(os.environ.get("AB_BATCH", "256")), 100         # AB_BATCH=1024: the two-utterances-per-workgroup kernel
    feats = torch.from_numpy(np.stack([synthetic_features(b, F) for b in range(B)])).cuda()
    out = torch.empty((B, F * 160), dtype=torch.int16, device="cuda")
    h = L.dss_lpcnet_batch_create(B, F)
    ms = []
    for it in range(6 if B <= 256 else 4):
        L.dss_lpcnet_batch_reset(h, -1)
        L.dss_lpcnet_batch_enable_timing(h, 1)
        assert L.dss_lpcnet_batch_synthesize_dev(h, feats.data_ptr(), B, F, 20, out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        ms.append(L.dss_lpcnet_batch_kernel_ms(h, 0))
    print(f"{os.path.basename(path)}: sample kernel ms {['%.2f' % m for m in ms]}  checksum {int(out.to(torch.int64).sum())}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for _ in range(2):
            for p in sys.argv[1:]:
                subprocess.check_call([sys.executable, __file__, "--child", os.path.abspath(p)])
