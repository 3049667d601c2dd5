"""Development aid: way points of the GRU B relay from the TIMED instantiation of the sample kernel.

    python tools/relay_stamps.py tools/ab/<variant>.so      (a build with -DDSS_RELAY_STAMP=1, tools/build_variant.sh)

Such a build accumulates the relay waves' clock stamps in scalar registers and leaves them in the first (silent) frame's PCM of
row 0; this script reads them back.  Cycles from barrier B, mean over the samples of the call."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "delayed-speech-synthesis_amd"))


def main(path):
    import numpy as np
    import torch
    from dss_amd.lpcnet_weights import synthetic_blob, synthetic_features
    L = C.CDLL(os.path.abspath(path), mode=C.RTLD_GLOBAL)
    vp, i = C.c_void_p, C.c_int
    L.dss_lpcnet_batch_create.restype = vp
    L.dss_lpcnet_batch_create.argtypes = [i, i]
    L.dss_lpcnet_batch_synthesize_dev.argtypes = [vp, vp, i, i, i, vp, vp]
    L.dss_lpcnet_load_model.argtypes = [C.c_char_p, C.c_size_t]
    blob = synthetic_blob(0)
    assert L.dss_lpcnet_load_model(blob, len(blob)) == 0
    B, F = int(os.environ.get("AB_BATCH", "256")), 100
    feats = torch.from_numpy(np.stack([synthetic_features(b, F) for b in range(B)])).cuda()
    out = torch.empty((B, F * 160), dtype=torch.int16, device="cuda")
    h = L.dss_lpcnet_batch_create(B, F)
    assert L.dss_lpcnet_batch_synthesize_dev(h, feats.data_ptr(), B, F, 20, out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    n = (F - 2) * 160
    st = out[0, :44].cpu().numpy().view(np.uint32).astype(np.float64) / n
    w7, w6, ra = st[:8], st[8:13], st[16:22]
    print(f"{os.path.basename(path)} ({B} rows): wave 6: segment 1 published {w6[0]:.0f}, segment 3 products {w6[1]:.0f}, has the sums {w6[2]:.0f}, "
          f"published {w6[3]:.0f}, speculation done {w6[4]:.0f} | wave 7: segment 2 products {w7[0]:.0f}, has the sums {w7[1]:.0f}, published {w7[2]:.0f}, "
          f"segment 4 first products {w7[7]:.0f}, all {w7[3]:.0f}, has the sums {w7[4]:.0f}, summed {w7[5]:.0f}, past barrier C {w7[6]:.0f} | "
          f"GRU A waves ready for barrier C at " + " ".join(f"{v:.0f}" for v in ra))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p) if len(sys.argv) == 2 else os.system(f"{sys.executable} {__file__} {p}")
