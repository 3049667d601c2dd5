"""The bidirectional recurrent decoder of the online path on the GPU, for many streams per launch.

``BiLstmDecoderGPU`` takes the weights of the reference's ``BidirectionalSpeechSynthesisModel`` (local/models.py:36-58; any
module with that ``state_dict``: ``lstm.weight_ih_l0`` ... ``lstm.bias_hh_l1_reverse``, ``regressor.weight``, ``regressor.bias``)
and maps the high-gamma frames of S streams to LPCNet features in three launches (``dss_dec_forward_dev``,
csrc/bilstm_decoder.hip) from the zero state, as ``DecodingModel.process`` calls the model (local/units.py:499-508).  Its
output goes straight into ``LPCNetBatch.synthesize_torch``."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_KEYS = tuple(f"lstm.{n}_l{layer}{rev}" for layer in (0, 1) for rev in ("", "_reverse")
              for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) + ("regressor.weight", "regressor.bias")


def fits(module) -> bool:
    """True when `module` is the reference's decoder as this kernel implements it: a 2-layer bidirectional LSTM with a linear
    head and exactly the reference's parameters (no dropout at inference, no projection), within the kernel's sizes."""
    try:
        sd = module.state_dict()
    except Exception:
        return False
    if set(sd.keys()) != set(_KEYS):
        return False
    h4, c = sd["lstm.weight_ih_l0"].shape
    h = h4 // 4
    o = sd["regressor.weight"].shape[0]
    ok = h4 == 4 * h and 1 <= h <= 128 and 1 <= c <= 256 and 1 <= o <= 32
    for rev in ("", "_reverse"):
        ok = ok and tuple(sd[f"lstm.weight_ih_l0{rev}"].shape) == (h4, c) and tuple(sd[f"lstm.weight_hh_l0{rev}"].shape) == (h4, h)
        ok = ok and tuple(sd[f"lstm.weight_ih_l1{rev}"].shape) == (h4, 2 * h) and tuple(sd[f"lstm.weight_hh_l1{rev}"].shape) == (h4, h)
    return bool(ok and tuple(sd["regressor.weight"].shape) == (o, 2 * h))


class BiLstmDecoderGPU:
    def __init__(self, max_streams: int, max_frames: int, module=None, state_dict=None):
        sd = state_dict if state_dict is not None else module.state_dict()
        w = [np.ascontiguousarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float32) for k in _KEYS]
        h4, c = w[0].shape
        self.S, self.T, self.C, self.H, self.O = int(max_streams), int(max_frames), int(c), int(h4 // 4), int(w[16].shape[0])
        self._L = _lib.require_gpu()
        self._h = self._L.dss_dec_create(self.S, self.T, self.C, self.H, self.O)
        if not self._h:
            raise MemoryError(self._L.dss_last_error().decode())
        ptrs = (C.c_void_p * 18)(*[a.ctypes.data for a in w])
        _lib.check(self._L.dss_dec_load_weights(self._h, ptrs))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.dss_dec_destroy(self._h)
            self._h = None

    def forward_torch(self, frames):
        """frames: CUDA (S, T, C) float64 or float32, S <= max_streams, T <= max_frames.  Returns float32 CUDA (S, T, n_outputs)."""
        import torch
        if frames.dtype not in (torch.float64, torch.float32):
            raise TypeError("frames must be float64 or float32")
        if frames.dim() != 3 or frames.shape[2] != self.C or not frames.is_cuda:
            raise ValueError(f"frames must be a CUDA tensor of shape (S, T, {self.C})")
        s, t = int(frames.shape[0]), int(frames.shape[1])
        if s < 1 or s > self.S or t < 1 or t > self.T:
            raise ValueError(f"{s} streams x {t} frames exceed this decoder's {self.S} x {self.T}")
        frames = frames.contiguous()
        feats = torch.empty((s, t, self.O), dtype=torch.float32, device=frames.device)
        _lib.check(self._L.dss_dec_forward_dev(self._h, frames.data_ptr(), int(frames.dtype == torch.float64), s, t, feats.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream))
        return feats

    __call__ = forward_torch
