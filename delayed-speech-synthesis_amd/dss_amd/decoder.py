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
    if any(str(getattr(v, "dtype", "")) not in ("torch.float32", "float32") for v in sd.values()):
        return False                                   # float64 / half weights: the module's own arithmetic, not this kernel's
    h4, c = sd["lstm.weight_ih_l0"].shape
    h = h4 // 4
    o = sd["regressor.weight"].shape[0]
    ok = h4 == 4 * h and 1 <= h <= 128 and 1 <= c <= 256 and 1 <= o <= 32
    for rev in ("", "_reverse"):
        ok = ok and tuple(sd[f"lstm.weight_ih_l0{rev}"].shape) == (h4, c) and tuple(sd[f"lstm.weight_hh_l0{rev}"].shape) == (h4, h)
        ok = ok and tuple(sd[f"lstm.weight_ih_l1{rev}"].shape) == (h4, 2 * h) and tuple(sd[f"lstm.weight_hh_l1{rev}"].shape) == (h4, h)
    return bool(ok and tuple(sd["regressor.weight"].shape) == (o, 2 * h))


def make_kernel(module, max_streams: int, max_frames: int, tol: float = 1e-4):
    """The kernel form of `module`, or None when the module has to run as it is.  Parameter names and shapes (``fits``) say what
    the module HOLDS, not what its ``forward`` DOES: a class with the reference's parameters and an activation, a clamp, a
    residual or an input normalisation around them would silently get the plain BiLSTM + Linear output.  So the kernel is kept
    only if, on a probe of 8 random frames from the zero state, it reproduces the module's own output within `tol` (the
    kernel's agreement with torch.nn.LSTM is ~1e-6); otherwise the module runs as given, with a warning."""
    if not fits(module):
        return None
    import torch
    import warnings
    k = BiLstmDecoderGPU(max_streams, max_frames, module)
    try:
        dev = next(module.parameters()).device
        x = torch.from_numpy(np.random.default_rng(20240229).standard_normal((1, min(8, k.T), k.C)).astype(np.float32)).to(dev)
        with torch.no_grad():
            want, _ = module(x, module.create_new_initial_state(batch_size=1, device=str(dev)))
        got = k(x.cuda())
        err = float((got - want.to(got.device)).abs().max())
    except Exception as e:                               # a module that cannot even be called like the reference's
        warnings.warn(f"decoder probe failed ({type(e).__name__}: {e}); running the module as given", RuntimeWarning, stacklevel=2)
        return None
    if not err <= tol:
        warnings.warn(f"decoder module has the reference's parameters but its forward differs from BiLSTM + Linear by {err:.3g} on a "
                      "probe; running the module as given (PyTorch-ROCm), not the kernel", RuntimeWarning, stacklevel=2)
        return None
    return k


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

    def forward_rows_torch(self, pool, in_rows, counts, feats, n_frames: int, stream=None):
        """Ragged form (``dss_dec_forward_rows_dev``): segment i has counts[i] frames in row in_rows[i] of ``pool`` (CUDA float32 or
        float64, (rows, row_frames, C)); features go to feats[i, :counts[i]] (CUDA float32 (>= n, n_frames, n_outputs), contiguous)."""
        import torch
        n = len(counts)
        c = np.ascontiguousarray(counts, dtype=np.int32)
        r = None if in_rows is None else np.ascontiguousarray(in_rows, dtype=np.int32)
        if pool.dim() != 3 or pool.shape[2] != self.C or not pool.is_cuda or not pool.is_contiguous():
            raise ValueError(f"pool must be a contiguous CUDA tensor (rows, row_frames, {self.C})")
        if pool.dtype not in (torch.float64, torch.float32):
            raise TypeError("pool must be float64 or float32")
        if feats.dtype != torch.float32 or not feats.is_contiguous() or feats.shape[0] < n or feats.shape[1] != n_frames or feats.shape[2] != self.O:
            raise ValueError(f"feats must be contiguous float32 (>= {n}, {n_frames}, {self.O})")
        if r is not None and (r.shape != (n,) or (n and int(r.max()) >= pool.shape[0])):
            raise ValueError("in_rows must name one pool row per segment")
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _lib.check(self._L.dss_dec_forward_rows_dev(self._h, pool.data_ptr(), int(pool.dtype == torch.float64), int(pool.shape[1]),
                                                    None if r is None else r.ctypes.data, c.ctypes.data, n, int(n_frames),
                                                    feats.data_ptr(), s))
        return feats
