"""ezmsg units of the online synthesis path, backed by libdss_hip.so.

Same unit classes, settings, stream names and message fields as the reference's local/units.py, so
``decode_online.py``'s imports (decode_online.py:10-16) and wiring (decode_online.py:149-164) resolve unchanged.
What differs is where the work happens:
  * HighGammaExtractor   -- band-pass + band-stop IIR cascades, warm-start frame buffer and windowed log power
                            run as ONE fused kernel per packet (reference: scipy sosfilt x2 + Cython, units.py:145-161)
  * DelayedLPCNetVocoder -- a whole decoded segment is synthesised by one launch of the persistent sample-rate
                            kernel (reference: one Python->C call per 10 ms frame, units.py:531-538)
  * the two LSTM units   -- unchanged PyTorch modules, on the GPU through PyTorch-ROCm
I/O units (ZMQ source, loggers, SoX sink) keep the reference's formats; they are not accelerated.
"""
from __future__ import annotations

import logging
import os
import struct
import sys
import time
from dataclasses import replace
from functools import reduce
from pathlib import Path
from typing import AsyncGenerator, Callable, Iterable, List, Optional

import numpy as np
import torch
import torch.nn as nn

from ._ez import TimeSeriesMessage, ez
from .common import SpeechSegmentHistory, VoiceActivityDetectionSmoothing

logger = logging.getLogger("units.py")

Transforms = Optional[List[Callable]]


class ClosedLoopMessage(TimeSeriesMessage):
    """TimeSeriesMessage plus the arrival time of the packet and the index of the segment's first frame."""
    received_at: Optional[float] = None
    previous_frames: Optional[float] = None


try:   # dataclass inheritance for the stand-in message type
    import dataclasses as _dc
    if _dc.is_dataclass(TimeSeriesMessage) and not ("received_at" in {f.name for f in _dc.fields(ClosedLoopMessage)}):
        ClosedLoopMessage = _dc.dataclass(ClosedLoopMessage)
except Exception:  # pragma: no cover
    pass


# ---------------------------------------------------------------------------------------------------------------
# amplifier source (BCI2000 over ZMQ PUB/SUB): wire format of development_amplifier.py:14-25
# ---------------------------------------------------------------------------------------------------------------
PACKET_HEADER = struct.Struct("=BBB HH")          # descriptor, supplement, dtype, n_channels, n_samples
PACKET_TOPIC = struct.Struct("=BBB").pack(4, 1, 2)


def interpret_bci2000_packet(data: bytes) -> np.ndarray:
    """7-byte header + float32 [n_channels][n_samples] -> float64 (n_samples, n_channels), C order."""
    _, _, _, n_channels, n_samples = PACKET_HEADER.unpack(data[:PACKET_HEADER.size])
    array = np.frombuffer(data[PACKET_HEADER.size:], dtype=np.float32).reshape(n_channels, n_samples)
    return np.transpose(array).astype(np.float64, order="C", copy=True)


class ZMQConnectorSettings(ez.Settings):
    fs: int
    port: int = 5556
    address: str = "localhost"


class ZMQConnectorState(ez.State):
    context = None
    socket = None
    header: struct.Struct = PACKET_HEADER
    topic: Optional[bytes] = None


class ZMQConnector(ez.Unit):
    SETTINGS: ZMQConnectorSettings
    STATE: ZMQConnectorState
    OUTPUT = ez.OutputStream(ClosedLoopMessage)

    def initialize(self) -> None:
        import zmq
        import zmq.asyncio
        self.STATE.topic = PACKET_TOPIC
        self.STATE.context = zmq.asyncio.Context()
        self.STATE.socket = self.STATE.context.socket(zmq.SUB)
        self.STATE.socket.setsockopt(zmq.RCVHWM, 1)          # drop stale packets rather than queue them
        self.STATE.socket.connect(f"tcp://{self.SETTINGS.address}:{self.SETTINGS.port}")
        self.STATE.socket.subscribe(self.STATE.topic)

    def shutdown(self) -> None:
        self.STATE.socket.unsubscribe(self.STATE.topic)
        self.STATE.socket.close()
        self.STATE.context.destroy()

    def interpret_bytes(self, data: bytes) -> np.ndarray:
        return interpret_bci2000_packet(data)

    @ez.publisher(OUTPUT)
    async def process(self) -> AsyncGenerator:
        while not self.STATE.socket.closed:
            data = self.interpret_bytes(await self.STATE.socket.recv())
            yield self.OUTPUT, ClosedLoopMessage(data=data, fs=self.SETTINGS.fs, received_at=time.time())


# ---------------------------------------------------------------------------------------------------------------
# feature extraction
# ---------------------------------------------------------------------------------------------------------------
class HighGammaExtractor:
    """70-170 Hz band-pass + 118-122 Hz band-stop (order-8 Butterworth SOS), 50 ms / 10 ms log-power frames.
    State (filter memories and the 40-sample frame overlap) lives on the GPU and carries across calls."""

    def __init__(self, fs, nb_electrodes, window_length=0.05, window_shift=0.01, l_freq: int = 70, h_freq: int = 170,
                 pre_transforms: Transforms = None, post_transforms: Transforms = None):
        from dss_amd.hga import HgaExtractorGPU, design_filters
        self.fs, self.nb_electrodes = fs, nb_electrodes
        self.window_length, self.window_shift = window_length, window_shift
        self.model_order, self.step_size = 4, 5
        self.pre_transform = self._compose_functions(*pre_transforms) if pre_transforms is not None else None
        self.post_transform = self._compose_functions(*post_transforms) if post_transforms is not None else None
        if not ((60 < l_freq < 120) or (120 < h_freq < 180)):
            logger.warning("l_freq and h_freq seem not to be in the recommended ranges!!")
        self.hg_filter, self.fh_filter, zi_hg, zi_fh = design_filters(fs, l_freq, h_freq, order=8)
        self._gpu = HgaExtractorGPU(1, nb_electrodes, fs=fs, window_length=window_length, window_shift=window_shift,
                                    filters=(self.hg_filter, self.fh_filter, zi_hg, zi_fh))
        # decode_online.py's pre-transform chain (reorder -> CAR -> select) is fused into the GPU front end
        self._fused_pre = None
        if pre_transforms is not None and len(pre_transforms) == 3 and hasattr(pre_transforms[0], "grid_mapping") \
                and hasattr(pre_transforms[1], "selection_masks_computation") \
                and hasattr(pre_transforms[2], "speech_grid_mapping") \
                and len(pre_transforms[2].speech_grid_mapping) == nb_electrodes:
            self._fused_pre = tuple(pre_transforms)

    @staticmethod
    def _compose_functions(*functions):
        return reduce(lambda f, g: lambda x: g(f(x)), functions, lambda x: x)

    def extract_features(self, data: np.ndarray):
        if self._fused_pre is not None:
            if getattr(self._gpu, "c_raw", None) != data.shape[1]:
                self._gpu.set_frontend_from_transforms(data.shape[1], *self._fused_pre)
            data = self._gpu.extract_raw(np.ascontiguousarray(data, dtype=np.float64))[0]
            return self.post_transform(data) if self.post_transform is not None else data
        if self.pre_transform is not None:
            data = self.pre_transform(data)
        data = self._gpu.extract(np.ascontiguousarray(data, dtype=np.float64))[0]
        if self.post_transform is not None:
            data = self.post_transform(data)
        return data


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
            pre_transforms=s.pre_transforms, post_transforms=s.post_transforms)

    @ez.publisher(OUTPUT)
    @ez.subscriber(INPUT)
    async def process(self, msg: TimeSeriesMessage) -> AsyncGenerator:
        features = self.STATE.hg_extractor.extract_features(msg.data)
        yield self.OUTPUT, replace(msg, data=features, fs=1 / self.SETTINGS.window_shift)


# ---------------------------------------------------------------------------------------------------------------
# loggers (file formats of the reference: raw little-endian dumps, Audacity label file, numbered wav files)
# ---------------------------------------------------------------------------------------------------------------
class LoggerSettings(ez.Settings):
    filename: str
    overwrite: bool
    config_filename: Optional[str] = None


class BinaryLoggerState(ez.State):
    file_descriptor = None
    shape: Optional[Iterable[int]] = None


def _open_log(filename: str, overwrite: bool, mode: str):
    filename = os.path.abspath(filename)
    os.makedirs(os.path.dirname(filename), exist_ok=True)
    if os.path.isfile(filename) and not overwrite:
        ext = os.path.basename(filename).split(".")[-1]
        raise PermissionError(f"The specified .{ext} file already exists and overwrite is disabled.")
    return open(filename, mode=mode)


class BinaryLogger(ez.Unit):
    """Appends ``message.data.tobytes()``; restore with np.fromfile(name, dtype).reshape(-1, columns)."""
    SETTINGS: LoggerSettings
    STATE: BinaryLoggerState
    INPUT = ez.InputStream(TimeSeriesMessage)

    def initialize(self) -> None:
        self.STATE.file_descriptor = _open_log(self.SETTINGS.filename, self.SETTINGS.overwrite, "wb")

    def shutdown(self) -> None:
        self.STATE.file_descriptor.flush()
        self.STATE.file_descriptor.close()

    @ez.subscriber(INPUT)
    async def write(self, message: TimeSeriesMessage) -> None:
        if self.STATE.shape is None:
            self.STATE.shape = list(message.data.shape)
            if len(self.STATE.shape) > 1:
                self.STATE.shape.pop(message.time_dim)
        self.STATE.file_descriptor.write(message.data.tobytes())


class VoiceActivityDetectionLoggerState(ez.State):
    file_descriptor = None


class VoiceActivityDetectionLogger(ez.Unit):
    """One tab-separated line per detected segment: start[s], stop[s], "<n> frames"."""
    SETTINGS: LoggerSettings
    STATE: VoiceActivityDetectionLoggerState
    INPUT = ez.InputStream(ClosedLoopMessage)

    def initialize(self) -> None:
        self.STATE.file_descriptor = _open_log(self.SETTINGS.filename, self.SETTINGS.overwrite, "w")

    def shutdown(self) -> None:
        self.STATE.file_descriptor.flush()
        self.STATE.file_descriptor.close()

    @ez.subscriber(INPUT)
    async def write(self, message: ClosedLoopMessage) -> None:
        start = message.previous_frames * 0.01
        stop = (message.previous_frames + len(message.data)) * 0.01
        self.STATE.file_descriptor.write(f"{start:.02f}\t{stop:.02f}\t{len(message.data)} frames\n")


class DelayedWavLoggerSettings(ez.Settings):
    base_path: Path
    overwrite: bool
    prefix: Optional[str] = None


class DelayedWavLoggerState(ez.State):
    speech_segment_counter: int = 1


class DelayedWavLogger(ez.Unit):
    SETTINGS: DelayedWavLoggerSettings
    STATE: DelayedWavLoggerState
    INPUT = ez.InputStream(TimeSeriesMessage)

    def initialize(self) -> None:
        os.makedirs(self.SETTINGS.base_path, exist_ok=True)
        if not hasattr(self.STATE, "speech_segment_counter") or self.STATE.speech_segment_counter is None:
            self.STATE.speech_segment_counter = 1

    @ez.subscriber(INPUT)
    async def write(self, message: TimeSeriesMessage) -> None:
        from scipy.io.wavfile import write as wavwrite
        prefix = self.SETTINGS.prefix if self.SETTINGS.prefix is not None else ""
        filename = os.path.join(Path(self.SETTINGS.base_path).as_posix(),
                                f"{prefix}_{self.STATE.speech_segment_counter:05d}.wav")
        self.STATE.speech_segment_counter += 1
        if not (os.path.isfile(filename) and not self.SETTINGS.overwrite):
            wavwrite(filename, 16000, message.data)


# ---------------------------------------------------------------------------------------------------------------
# neural VAD gate and decoder (PyTorch-ROCm)
# ---------------------------------------------------------------------------------------------------------------
class FilterSpeechSegmentsSettings(ez.Settings):
    nb_features: int
    fs: int
    vad_architecture: type
    buffer_size: int
    context_frames: int = 0
    vad_weights_path: Optional[Path] = None
    vad_parameters: Optional[dict] = None


class FilterSpeechSegmentsState(ez.State):
    device: str = "cpu"
    history: Optional[SpeechSegmentHistory] = None
    smoothing: Optional[VoiceActivityDetectionSmoothing] = None
    vad_model: Optional[nn.Module] = None
    vad_state = None
    frame_counter: int = 0


class FilterSpeechSegments(ez.Unit):
    SETTINGS: FilterSpeechSegmentsSettings
    STATE: FilterSpeechSegmentsState
    INPUT = ez.InputStream(ez.Message)
    OUTPUT = ez.OutputStream(ez.Message)

    def initialize(self) -> None:
        s, st = self.SETTINGS, self.STATE
        st.device = "cuda" if torch.cuda.is_available() else "cpu"
        st.history = SpeechSegmentHistory(nb_features=s.nb_features, buffer_size=s.buffer_size, context=s.context_frames)
        st.smoothing = VoiceActivityDetectionSmoothing(nb_features=s.nb_features, context_frames=5)
        params = s.vad_parameters if s.vad_parameters is not None else dict()
        st.vad_model = s.vad_architecture(**params).to(st.device)
        if s.vad_weights_path is not None:
            st.vad_model.load_state_dict(torch.load(Path(s.vad_weights_path).as_posix(), map_location=st.device))
        st.vad_state = st.vad_model.create_new_initial_state(batch_size=1, device=st.device)
        st.vad_model.eval()
        st.frame_counter = 0

    @ez.publisher(OUTPUT)
    @ez.subscriber(INPUT)
    async def process(self, msg: ClosedLoopMessage) -> AsyncGenerator:
        st = self.STATE
        frames = torch.from_numpy(np.expand_dims(msg.data, 0)).float().to(st.device)      # batch x time x features
        with torch.no_grad():
            predictions, st.vad_state = st.vad_model(frames, st.vad_state)
        predictions = torch.argmax(predictions, dim=2).flatten().detach().cpu().numpy()
        data, predictions = st.smoothing.insert(data=msg.data, speech_labels=predictions)
        segments = st.history.insert(data=data, speech_labels=predictions)
        st.frame_counter += len(msg.data)
        for segment in segments:
            previous = st.frame_counter - len(segment) - (len(msg.data) - np.count_nonzero(predictions))
            yield self.OUTPUT, replace(msg, data=segment, fs=100, previous_frames=previous)


class RecurrentNeuralDecodingModelSettings(ez.Settings):
    path_to_model_weights: Optional[str]
    model: type
    params: Optional[dict]
    config_filename: Optional[str] = None


class RecurrentNeuralDecodingModelState(ez.State):
    decoding_model: Optional[nn.Module] = None
    device: Optional[str] = None
    H = None


class RecurrentNeuralDecodingModel(ez.Unit):
    SETTINGS: RecurrentNeuralDecodingModelSettings
    STATE: RecurrentNeuralDecodingModelState
    INPUT = ez.InputStream(TimeSeriesMessage)
    OUTPUT = ez.OutputStream(TimeSeriesMessage)

    def initialize(self) -> None:
        s, st = self.SETTINGS, self.STATE
        params = s.params if s.params is not None else dict()
        st.device = "cuda" if torch.cuda.is_available() else "cpu"
        st.decoding_model = s.model(**params).to(st.device)
        if s.path_to_model_weights is not None:
            st.decoding_model.load_state_dict(torch.load(s.path_to_model_weights, map_location=st.device))
        st.decoding_model.eval()
        st.H = st.decoding_model.create_new_initial_state(batch_size=1, device=st.device)

    @ez.subscriber(INPUT)
    @ez.publisher(OUTPUT)
    async def decode(self, msg: TimeSeriesMessage) -> AsyncGenerator:
        st = self.STATE
        frames = torch.from_numpy(np.expand_dims(msg.data, 0)).float().to(st.device)
        with torch.no_grad():
            predictions, st.H = st.decoding_model(frames, st.H)
        predictions = np.squeeze(predictions.detach().cpu().numpy(), axis=0)
        st.H = st.decoding_model.create_new_initial_state(batch_size=1, device=st.device)   # fresh state per segment
        yield self.OUTPUT, replace(msg, data=predictions, fs=100)


# ---------------------------------------------------------------------------------------------------------------
# vocoder + sink
# ---------------------------------------------------------------------------------------------------------------
class LPCNetState(ez.State):
    lpcnet = None


class DelayedLPCNetVocoder(ez.Unit):
    """(L, 20) decoded LPCNet features -> int16[L*160] at 16 kHz.  One decoder state for the unit's lifetime,
    carried across segments exactly like the reference's single ``LPCNet.LPCNet()`` instance (units.py:524)."""
    STATE: LPCNetState
    INPUT = ez.InputStream(TimeSeriesMessage)
    OUTPUT = ez.OutputStream(TimeSeriesMessage)
    MAX_SEGMENT_FRAMES = 2200            # FilterSpeechSegments' ring buffer holds 2000 frames (decode_online.py:116)

    def initialize(self) -> None:
        from dss_amd.lpcnet import LPCNetBatch
        self.STATE.lpcnet = LPCNetBatch(1, self.MAX_SEGMENT_FRAMES)

    def shutdown(self) -> None:
        self.STATE.lpcnet.close()
        self.STATE.lpcnet = None

    @ez.subscriber(INPUT)
    @ez.publisher(OUTPUT)
    async def synthesize(self, msg: TimeSeriesMessage) -> AsyncGenerator:
        features = np.ascontiguousarray(msg.data.astype(np.float32))
        chunks = [self.STATE.lpcnet.synthesize(features[None, a:a + self.MAX_SEGMENT_FRAMES])[0]
                  for a in range(0, len(features), self.MAX_SEGMENT_FRAMES)]
        yield self.OUTPUT, replace(msg, data=np.hstack(chunks), fs=16000)


class DelayedStdoutForSoX(ez.Unit):
    """Raw s16le to stdout for ``play -t raw -r 16000 -e signed -b 16 -c 1`` (replicate.sh:115-116)."""
    INPUT = ez.InputStream(ClosedLoopMessage)

    @ez.subscriber(INPUT)
    async def print(self, msg: ClosedLoopMessage) -> None:
        sys.stdout.buffer.write(msg.data.tobytes())
        sys.stdout.flush()
