"""The units of the online graph whose work moves to the GPU, with the surface of the
reference's local/units.py classes of the same names (settings, stream names, message fields, handler names).

  HighGammaExtractor   (units.py:97-161)   IIR x2 + warm-start frame buffer + log power as fused kernels per packet;
                                           decode_online.py's three pre-transforms collapse into the GPU front end
  HighGammaActivity    (units.py:185-207)  unit wrapper, msg.fs = 1 / window_shift
  DelayedLPCNetVocoder (units.py:517-538)  a whole decoded segment per launch of the persistent sample-rate kernel;
                                           one decoder state for the unit's lifetime, as units.py:524
  gpu_decoding_unit(user's RecurrentNeuralDecodingModel) (units.py:472-508)  a SUBCLASS of the user's own unit: its
                                           initialize() stays the user's, decode() runs the reference's decoder architecture
                                           on the library's kernels (three launches per segment), any other module as it was

Everything else of the reference's unit module (ZMQ source, loggers, VAD gate, SoX sink) is I/O or stock
PyTorch and stays the user's own code: ``python -m dss_amd.run decode_online.py ...`` swaps exactly the classes above
into the user's ``local.units`` and runs the script unchanged (dss_amd/run.py).  Without ezmsg installed the
classes are built on a small stand-in (dss_amd/_ez.py) so they can be constructed and driven directly.
"""
from __future__ import annotations

import logging
from dataclasses import replace
from typing import AsyncGenerator, Callable, List, Optional

import numpy as np

from ._ez import TimeSeriesMessage, ez

logger = logging.getLogger("dss_amd.units")
Transforms = Optional[List[Callable]]


class ClosedLoopMessage(TimeSeriesMessage):
    """Message type of the reference graph (units.py:29-35): data, fs, plus arrival time / first-frame index."""
    received_at: Optional[float] = None
    previous_frames: Optional[float] = None


import dataclasses as _dc                                  # the stand-in message type is a plain dataclass
if _dc.is_dataclass(TimeSeriesMessage) and "received_at" not in {f.name for f in _dc.fields(ClosedLoopMessage)}:
    ClosedLoopMessage = _dc.dataclass(ClosedLoopMessage)


def _chain(functions):
    def run(x):
        for f in functions:
            x = f(x)
        return x
    return run


class HighGammaExtractor:
    """70-170 Hz band-pass + 118-122 Hz band-stop (order-8 Butterworth SOS), 50 ms / 10 ms log-power frames
    (units.py:102-161).  Filter memories and the 40-sample frame overlap live on the GPU and carry across calls."""

    def __init__(self, fs, nb_electrodes, window_length=0.05, window_shift=0.01, l_freq: int = 70, h_freq: int = 170,
                 pre_transforms: Transforms = None, post_transforms: Transforms = None):
        from .hga import HgaExtractorGPU, reference_filters
        self.fs, self.nb_electrodes = fs, nb_electrodes
        self.window_length, self.window_shift = window_length, window_shift
        self.pre_transform = _chain(pre_transforms) if pre_transforms is not None else None
        self.post_transform = _chain(post_transforms) if post_transforms is not None else None
        if not ((60 < l_freq < 120) or (120 < h_freq < 180)):
            logger.warning("band edges %s-%s Hz lie outside the high-gamma range the reference recommends", l_freq, h_freq)
        self.hg_filter, self.fh_filter, zi_hg, zi_fh = reference_filters(fs, l_freq, h_freq)
        self._gpu = HgaExtractorGPU(1, nb_electrodes, fs=fs, window_length=window_length, window_shift=window_shift,
                                    filters=(self.hg_filter, self.fh_filter, zi_hg, zi_fh))
        # decode_online.py:65-85's chain (reorder -> per-grid CAR -> select) is recognised by the attributes its three
        # objects expose and fused into the GPU front end; any other chain runs on the host as given
        self._fused_pre = None
        p = pre_transforms
        if p is not None and len(p) == 3 and hasattr(p[0], "grid_mapping") and hasattr(p[1], "selection_masks_computation") \
                and hasattr(p[2], "speech_grid_mapping") and len(p[2].speech_grid_mapping) == nb_electrodes:
            self._fused_pre = tuple(p)
        # decode_online.py:88-97's single post-transform, ZScoreNormalization (common.py:367-376), is applied by the library
        # right behind the log (dss_hga_set_zscore): (frames - means) / stds, the same two IEEE operations numpy performs.
        # It is recognised by what it DOES, not by its attribute names: the object is run on a probe array and must return,
        # bit for bit, (x - channel_means) / channel_stds -- a look-alike that clips, adds an epsilon or casts keeps its own
        # __call__ on the host, like any other chain.
        self._fused_post = False
        q = post_transforms
        if q is not None and len(q) == 1 and hasattr(q[0], "channel_means") and hasattr(q[0], "channel_stds"):
            m = np.asarray(q[0].channel_means, dtype=np.float64).reshape(-1)
            sd = np.asarray(q[0].channel_stds, dtype=np.float64).reshape(-1)
            if m.shape == (nb_electrodes,) and sd.shape == (nb_electrodes,) and self._is_plain_zscore(q[0], m, sd):
                self._gpu.set_zscore(m, sd)
                self._fused_post = True

    @staticmethod
    def _is_plain_zscore(obj, m, sd) -> bool:
        probe = np.random.default_rng(12345).standard_normal((7, m.shape[0])) * 3.0 + m
        probe[0] = m                                           # zeros, huge and tiny values: what a clip or an epsilon would bend
        probe[1] = m + 1e12 * sd
        probe[2] = m - 1e-12 * sd
        try:
            got = np.asarray(obj(probe.copy()))
        except Exception:
            return False
        want = (probe - m) / sd
        return got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want)

    def extract_features(self, data: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.float64)
        if self._fused_pre is not None:
            if getattr(self._gpu, "c_raw", None) != data.shape[1]:
                self._gpu.set_frontend_from_transforms(data.shape[1], *self._fused_pre)
            out = self._gpu.extract_raw(data)[0]
        else:
            if self.pre_transform is not None:
                data = np.ascontiguousarray(self.pre_transform(data), dtype=np.float64)
            out = self._gpu.extract(data)[0]
        if self._fused_post or self.post_transform is None:
            return out
        return self.post_transform(out)


class HighGammaActivitySettings(ez.Settings):
    fs: int
    nb_electrodes: int
    window_length: float = 0.05
    window_shift: float = 0.01
    l_freq: int = 70
    h_freq: int = 170
    pre_transforms: Transforms = None
    post_transforms: Transforms = None


class HighGammaActivityState(ez.State):
    hg_extractor: Optional[HighGammaExtractor] = None


class HighGammaActivity(ez.Unit):
    SETTINGS: HighGammaActivitySettings
    STATE: HighGammaActivityState
    INPUT = ez.InputStream(TimeSeriesMessage)
    OUTPUT = ez.OutputStream(TimeSeriesMessage)

    def initialize(self) -> None:
        s = self.SETTINGS
        self.STATE.hg_extractor = HighGammaExtractor(
            fs=s.fs, nb_electrodes=s.nb_electrodes, window_length=s.window_length, window_shift=s.window_shift,
            l_freq=s.l_freq, h_freq=s.h_freq, pre_transforms=s.pre_transforms, post_transforms=s.post_transforms)

    @ez.publisher(OUTPUT)
    @ez.subscriber(INPUT)
    async def process(self, msg: TimeSeriesMessage) -> AsyncGenerator:
        frames = self.STATE.hg_extractor.extract_features(msg.data)
        yield self.OUTPUT, replace(msg, data=frames, fs=1 / self.SETTINGS.window_shift)


def gpu_decoding_unit(base):
    """The user's own ``RecurrentNeuralDecodingModel`` (units.py:472-508) with its forward pass on the library's kernels.

    ``python -m dss_amd.run`` calls this with the class it finds in the user's ``local.units``: the subclass keeps the
    settings, state, streams and -- by calling it -- the ``initialize()`` of the user's class (building the module, loading the
    state_dict, ``eval()``, the initial state all stay the user's code), then looks at the module that came out: the
    reference's architecture (2-layer bidirectional LSTM + linear head, models.py:36-58, checked by parameter shapes AND a
    probe of its ``forward``) gets ``decode`` in three launches of csrc/bilstm_decoder.hip from the same weights; any other
    module -- or no GPU -- is called exactly as the base class calls it."""
    import inspect

    class RecurrentNeuralDecodingModel(base):
        __doc__ = gpu_decoding_unit.__doc__
        MAX_SEGMENT_FRAMES = 2200            # FilterSpeechSegments' ring holds 2000 frames (decode_online.py:116)

        def initialize(self) -> None:
            r = super().initialize()
            st = self.STATE
            st.kernel = None
            if str(getattr(st, "device", "cpu")).startswith("cuda") and getattr(st, "decoding_model", None) is not None:
                from . import decoder as _dec
                st.kernel = _dec.make_kernel(st.decoding_model, 1, self.MAX_SEGMENT_FRAMES)
            return r

        @ez.subscriber(getattr(base, "INPUT", None))
        @ez.publisher(getattr(base, "OUTPUT", None))
        async def decode(self, msg: TimeSeriesMessage) -> AsyncGenerator:
            k = getattr(self.STATE, "kernel", None)
            n = int(np.shape(msg.data)[0])
            if k is None or not (1 <= n <= k.T):
                async for item in super().decode(msg):          # the user's own handler, as it is
                    yield item
                return
            import torch
            x = torch.from_numpy(np.expand_dims(msg.data, 0)).float().cuda()
            yield self.OUTPUT, replace(msg, data=k(x)[0].cpu().numpy(), fs=100)

    RecurrentNeuralDecodingModel.__module__ = __name__
    RecurrentNeuralDecodingModel.__qualname__ = "RecurrentNeuralDecodingModel"
    if not inspect.isclass(base):
        raise TypeError("gpu_decoding_unit needs the user's unit CLASS")
    return RecurrentNeuralDecodingModel


from ._standin_units import (RecurrentNeuralDecodingModelSettings, RecurrentNeuralDecodingModelState,   # noqa: E402
                             StandInDecodingUnit)

# Without the user's tree (tests, this image: no ezmsg / zmq / mne, so the reference's local/units.py cannot be imported) the
# factory is applied to a stand-in of the user's class; `python -m dss_amd.run` never uses this name, it wraps the real one.
RecurrentNeuralDecodingModel = gpu_decoding_unit(StandInDecodingUnit)


class LPCNetState(ez.State):
    lpcnet = None


class DelayedLPCNetVocoder(ez.Unit):
    """(L, 20) decoded LPCNet features -> int16[L*160] at 16 kHz (units.py:531-538)."""
    STATE: LPCNetState
    INPUT = ez.InputStream(TimeSeriesMessage)
    OUTPUT = ez.OutputStream(TimeSeriesMessage)
    MAX_SEGMENT_FRAMES = 2200            # FilterSpeechSegments' ring holds 2000 frames (decode_online.py:116)

    def initialize(self) -> None:
        from .lpcnet import LPCNetBatch
        self.STATE.lpcnet = LPCNetBatch(1, self.MAX_SEGMENT_FRAMES)

    def shutdown(self) -> None:
        if self.STATE.lpcnet is not None:
            self.STATE.lpcnet.close()
            self.STATE.lpcnet = None

    @ez.subscriber(INPUT)
    @ez.publisher(OUTPUT)
    async def synthesize(self, msg: TimeSeriesMessage) -> AsyncGenerator:
        feats = np.ascontiguousarray(msg.data, dtype=np.float32)
        step = self.MAX_SEGMENT_FRAMES
        pcm = [self.STATE.lpcnet.synthesize(feats[None, a:a + step])[0] for a in range(0, len(feats), step)]
        yield self.OUTPUT, replace(msg, data=np.concatenate(pcm) if pcm else np.zeros(0, np.int16), fs=16000)
