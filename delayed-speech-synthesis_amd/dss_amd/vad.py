"""The neural voice-activity detector of the online path on the GPU, for many streams per launch.

``VadLstmGPU`` takes the weights of the reference's ``UnidirectionalVoiceActivityDetector`` (local/models.py:11-33; any
module with that ``state_dict``: ``lstm.weight_ih_l0`` ... ``classifier.bias``, two layers) and steps S streams by the
frames of one packet in ONE launch (``dss_vad_step_dev``, csrc/vad_lstm.hip), carrying (h, c) across packets like
``FilterSpeechSegments`` does (local/units.py:432-434).  Its labels go straight into ``SpeechGateGPU.push_torch``."""
from __future__ import annotations

import numpy as np

from . import _lib

_KEYS = ("lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
         "lstm.weight_ih_l1", "lstm.weight_hh_l1", "lstm.bias_ih_l1", "lstm.bias_hh_l1",
         "classifier.weight", "classifier.bias")


def fits(module) -> bool:
    """True when `module` is the reference's detector as this kernel implements it: a 2-layer unidirectional LSTM with a
    2-class linear head and exactly the reference's parameters (no dropout at inference, no projection)."""
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
    return (h4 == 4 * h and h <= 160 and c <= 128 and tuple(sd["lstm.weight_hh_l0"].shape) == (h4, h)
            and tuple(sd["lstm.weight_ih_l1"].shape) == (h4, h) and tuple(sd["classifier.weight"].shape) == (2, h))


def make_kernel(module, n_streams: int, tol: float = 1e-4):
    """The kernel form of `module`, or None when the module has to run as it is: ``fits`` reads parameter names and shapes, this
    also checks what the module's ``forward`` does -- its logits on 8 random frames from the zero state must be the kernel's
    within `tol` (see dss_amd.decoder.make_kernel)."""
    if not fits(module):
        return None
    import torch
    import warnings
    k = VadLstmGPU(n_streams, module)
    try:
        dev = next(module.parameters()).device
        x = np.random.default_rng(20240301).standard_normal((1, 8, k.C)).astype(np.float32)
        with torch.no_grad():
            want, _ = module(torch.from_numpy(x).to(dev), module.create_new_initial_state(batch_size=1, device=str(dev)))
        probe = VadLstmGPU(1, module)
        _, got = probe.step_torch(torch.from_numpy(x).cuda(), want_logits=True)
        err = float((got - want.to(got.device)).abs().max())
    except Exception as e:
        warnings.warn(f"detector probe failed ({type(e).__name__}: {e}); running the module as given", RuntimeWarning, stacklevel=2)
        return None
    if not err <= tol:
        warnings.warn(f"detector module has the reference's parameters but its forward differs from LSTM + Linear by {err:.3g} on a "
                      "probe; running the module as given (PyTorch-ROCm), not the kernel", RuntimeWarning, stacklevel=2)
        return None
    return k


class VadLstmGPU:
    def __init__(self, n_streams: int, module=None, state_dict=None):
        sd = state_dict if state_dict is not None else module.state_dict()
        w = [np.ascontiguousarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k], dtype=np.float32) for k in _KEYS]
        h4, c = w[0].shape
        self.S, self.C, self.H = int(n_streams), int(c), int(h4 // 4)
        self._L = _lib.require_gpu()
        self._h = self._L.dss_vad_create(self.S, self.C, self.H)
        if not self._h:
            raise MemoryError(self._L.dss_last_error().decode())
        _lib.check(self._L.dss_vad_load_weights(self._h, *[a.ctypes.data for a in w]))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.dss_vad_destroy(self._h)
            self._h = None

    def reset(self, stream: int = -1, hip_stream=None):
        """Zero state of one stream (or all).  Enqueued on the stream the steps run on (torch's current one by default), so it is
        ordered against steps in flight whatever kind of stream that is."""
        import torch
        s = torch.cuda.current_stream().cuda_stream if hip_stream is None else hip_stream
        _lib.check(self._L.dss_vad_reset_async(self._h, int(stream), s))

    def step_torch(self, frames, want_logits: bool = False):
        """frames: CUDA (S, W, C) float64 or float32.  Returns int32 CUDA labels (S, W) [, float32 logits (S, W, 2)]."""
        import torch
        if frames.dtype not in (torch.float64, torch.float32):
            raise TypeError("frames must be float64 or float32")
        if frames.dim() != 3 or frames.shape[0] != self.S or frames.shape[2] != self.C or not frames.is_cuda:
            raise ValueError(f"frames must be a CUDA tensor of shape ({self.S}, W, {self.C})")
        frames = frames.contiguous()
        w = frames.shape[1]
        labels = torch.empty((self.S, w), dtype=torch.int32, device=frames.device)
        logits = torch.empty((self.S, w, 2), dtype=torch.float32, device=frames.device) if want_logits else None
        _lib.check(self._L.dss_vad_step_dev(self._h, frames.data_ptr(), int(frames.dtype == torch.float64), w, labels.data_ptr(),
                                            logits.data_ptr() if want_logits else None, torch.cuda.current_stream().cuda_stream))
        return (labels, logits) if want_logits else labels

    def state(self):
        """(h, c), host float32 arrays [2][S][H]."""
        h = np.empty((2, self.S, self.H), np.float32)
        c = np.empty((2, self.S, self.H), np.float32)
        _lib.check(self._L.dss_vad_state(self._h, h.ctypes.data, c.ctypes.data, 0))
        return h, c
