"""Device-resident ECoG -> audio pipelines built from the hot-path operators (BASELINE.json configs 3 and 5).

``SegmentPipeline``   config 3: a batch of 64-channel ECoG segments -> HGA frames -> z-score -> BiLSTM decoder
                      (PyTorch-ROCm, reference local/models.py) -> LPCNet -> 16 kHz PCM.  Nothing leaves HBM
                      between the stages.
``GatedStreamingPipeline`` config 5 with the reference's segment gating (decode_online.py): the neural VAD runs on
                      all streams' frames at once, the smoothing and segment ring buffers live on the device
                      (csrc/speech_gate.hip), and only completed speech segments are decoded (whole-segment BiLSTM,
                      as the reference does) and synthesised -- one ragged LPCNet launch per tick for the streams
                      whose segment just closed, each continuing that stream's vocoder state.
``StreamingPipeline`` config 5: S concurrent streams advanced one amplifier packet (40 samples = 4 frames) at a
                      time; HGA filter state, frame overlap and LPCNet decoder state persist per stream.  The
                      bidirectional decoder is whole-segment in the reference (units.py:499-508); here it runs on
                      each packet's frames with a fresh state, i.e. VAD gating disabled and every frame decoded
                      (SURVEY.md 8d, config 5).
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch

from .gate import SpeechGateGPU
from .hga import HgaExtractorGPU
from .lpcnet import FRAME_SIZE, LPCNetBatch


class _DecoderMixin:
    def _make_decoder(self, n_channels, decoder, seed):
        if decoder is None:
            from .models import BidirectionalSpeechSynthesisModel
            torch.manual_seed(seed)         # no trained checkpoint exists offline: seeded random weights
            decoder = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=n_channels)
        return decoder.eval().cuda()

    def _make_decoder_kernel(self, max_streams, max_frames, use_kernel=True):
        """The reference's decoder (2-layer bidirectional LSTM + linear head, models.py:36-58) runs in three launches on the
        library's own kernels (dss_dec_forward_dev: all streams, all frames of the call, zero initial state); any other module
        is called as given, on PyTorch-ROCm."""
        from . import decoder as _dec
        self.dec_gpu = _dec.make_kernel(self.decoder, max_streams, max_frames) if use_kernel else None       # fits() + a probe of forward()

    def _decode(self, x):
        """x: CUDA (B, T, C) float64 or float32 -> float32 (B, T, 20), from a fresh zero state (units.py:499-508)."""
        if self.dec_gpu is not None and x.shape[0] <= self.dec_gpu.S and x.shape[1] <= self.dec_gpu.T:
            return self.dec_gpu(x)
        feats, _ = self.decoder(x.to(torch.float32), self.decoder.create_new_initial_state(batch_size=x.shape[0], device="cuda"))
        return feats.contiguous()


class SegmentPipeline(_DecoderMixin):
    def __init__(self, batch: int, n_samples: int = 1040, n_channels: int = 64, fs: int = 1000,
                 channel_means: Optional[np.ndarray] = None, channel_stds: Optional[np.ndarray] = None,
                 decoder: Optional[torch.nn.Module] = None, seed: int = 0, window_length: float = 0.05,
                 window_shift: float = 0.01, use_decoder_kernel: bool = True):
        self.B, self.n, self.C = batch, n_samples, n_channels
        self.hga = HgaExtractorGPU(batch, n_channels, fs=fs, window_length=window_length, window_shift=window_shift)
        self.frames = self.hga.frames_for(n_samples)
        self.decoder = self._make_decoder(n_channels, decoder, seed)
        self._make_decoder_kernel(batch, self.frames, use_decoder_kernel)
        self.vocoder = LPCNetBatch(batch, self.frames)
        mean = np.zeros(n_channels) if channel_means is None else np.asarray(channel_means, dtype=np.float64)
        std = np.ones(n_channels) if channel_stds is None else np.asarray(channel_stds, dtype=np.float64)
        self.mean = torch.from_numpy(mean).cuda()
        self.std = torch.from_numpy(std).cuda()
        self._zs = (mean, std)
        # ZScoreNormalization as the epilogue of the HGA launch (hga_fused_kernel's, or hga_window_kernel's when the window
        # shape sends the extractor to its three-launch form).  Uploaded once; the intermediates tap below only switches it
        # off and on (the library keeps the device copies: no allocation, no copy, no synchronisation per segment).
        self._zs_in_kernel = True
        self._zs_on = True
        self.hga.set_zscore(mean, std)

    @torch.no_grad()
    def __call__(self, ecog: torch.Tensor, return_intermediates: bool = False):
        """ecog: CUDA float64 (B, n_samples, C).  Returns int16 CUDA (B, frames*160)."""
        self.hga.reset()                                   # a fresh extractor per segment (prepare_corpus.py:147-176)
        self.vocoder.reset_async()                         # a fresh decoder per segment (training.py:193)
        if self._zs_in_kernel and not return_intermediates:
            if not self._zs_on:
                self.hga.set_zscore(*self._zs)
                self._zs_on = True
            hga = None
            z = self.hga.extract_torch(ecog, apply_log=True)                      # z-scored frames straight from the launch
        else:                                                                 # (test tap: the frames before the z-score)
            if self._zs_on:
                self.hga.set_zscore(None)
                self._zs_on = False
            hga = self.hga.extract_torch(ecog, apply_log=True)                # (B, W, C) float64
            z = (hga - self.mean) / self.std                                  # ZScoreNormalization (.float() in the decoder call)
        feats = self._decode(z)
        pcm = self.vocoder.synthesize_torch(feats)
        return (pcm, hga, feats) if return_intermediates else pcm


class StreamingPipeline(_DecoderMixin):
    def __init__(self, n_streams: int, n_channels: int = 64, fs: int = 1000, packet: int = 40,
                 decoder: Optional[torch.nn.Module] = None, seed: int = 0, use_graph: bool = True,
                 use_decoder_kernel: bool = True):
        self.S, self.C, self.packet = n_streams, n_channels, packet
        self.hga = HgaExtractorGPU(n_streams, n_channels, fs=fs)
        self.decoder = self._make_decoder(n_channels, decoder, seed)
        self._make_decoder_kernel(n_streams, 8, use_decoder_kernel)
        self.vocoder = LPCNetBatch(n_streams, 8)
        self._in = torch.empty((n_streams, packet, n_channels), dtype=torch.float64, device="cuda")
        # The steady-state tick (every packet after the first: 4 frames) is a fixed sequence of small launches: HGA, the decoder
        # (three launches of the library's own kernels; ~20 of MIOpen's for a module of another architecture), the frame-rate
        # network, the sample-rate kernel.  It is captured once into a HIP graph and
        # replayed per packet, which takes the per-launch host cost off the latency path.  Same kernels, same results.
        self.use_graph = use_graph
        self._graph = None
        self._graph_out = None
        self._graph_failed = False
        self.hga_first = True            # the first packet is the frame buffer's CASE 2 (one frame): not the captured shape
        self.last_hga = None             # test taps: the last tick's HGA frames (S, W, C) f64 and decoder output (S, W, 20) f32
        self.last_feats = None           #   (device tensors; in graph mode the captured tick's own buffers, rewritten by every replay)

    @torch.no_grad()
    def _tick(self):
        hga = self.hga.extract_torch(self._in, apply_log=True)
        feats = self._decode(hga)
        self.last_hga, self.last_feats = hga, feats
        return self.vocoder.synthesize_torch(feats)

    def _capture(self):
        """Capture one steady-state tick.  Capturing enqueues nothing, so the decoder states are untouched."""
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._tick()
            self._graph, self._graph_out = g, out
            self._graph_taps = (self.last_hga, self.last_feats)
        except Exception as e:      # e.g. an RNN backend that cannot be captured: stay on eager launches
            self._graph_failed = True
            torch.cuda.synchronize()
            import warnings
            warnings.warn(f"streaming tick could not be captured into a HIP graph ({type(e).__name__}: {e}); using eager launches",
                          RuntimeWarning)

    @torch.no_grad()
    def push(self, packets: np.ndarray) -> np.ndarray:
        """packets: host float64 (S, packet, C), one amplifier packet per stream.  Returns host int16 (S, W*160)
        for the W frames this packet completed (1 for the very first packet, then 4)."""
        # plain pageable copies: CPU access to pinned (fine-grained) host memory is far slower than the copy itself
        self._in.copy_(torch.from_numpy(np.ascontiguousarray(packets, dtype=np.float64)))
        W = self.hga.frames_for(self.packet)
        steady = W * FRAME_SIZE == 4 * FRAME_SIZE and not self.hga_first
        if self.use_graph and steady and not self._graph_failed:
            if self._graph is None:
                self._capture()
            if self._graph is not None:
                self._graph.replay()
                self.last_hga, self.last_feats = self._graph_taps
                return self._graph_out.cpu().numpy()
        pcm = self._tick()
        self.hga_first = False
        assert pcm.shape[1] == W * FRAME_SIZE
        return pcm.cpu().numpy()

    def measure_latency(self, n_packets: int = 250, seed: int = 0):
        """Packet-in (host) -> PCM-out (host) latency over n_packets ticks; returns milliseconds per tick."""
        rng = np.random.default_rng(seed)
        lat = []
        for _ in range(n_packets):
            pk = rng.standard_normal((self.S, self.packet, self.C)) * 50.0
            t0 = time.perf_counter()
            self.push(pk)
            lat.append((time.perf_counter() - t0) * 1e3)
        return np.asarray(lat)


class GatedStreamingPipeline(_DecoderMixin):
    """S streams through HighGammaActivity -> FilterSpeechSegments -> RecurrentNeuralDecodingModel ->
    DelayedLPCNetVocoder (decode_online.py:115-135), all streams per tick in batched launches.

    The tick (extractor, detector, gate: three launches for all streams) never waits for a vocoder.  Segments that close
    on a tick are handed to a ``SegmentSynthesisQueue`` (side streams: whole-segment decoder + ragged vocoder launch + PCM
    copy per job) and ``push()`` returns the segments FINISHED since the last tick.  The reference's own behaviour -- its
    one stream blocks while it vocodes (units.py:531-538) -- is ``asynchronous=False``: every tick then waits for what it
    started, and returns it."""

    def __init__(self, n_streams: int, n_channels: int = 64, fs: int = 1000, packet: int = 40,
                 buffer_size: int = 2000, context_frames: int = 50, smoothing_context: int = 5,
                 channel_means: Optional[np.ndarray] = None, channel_stds: Optional[np.ndarray] = None,
                 decoder: Optional[torch.nn.Module] = None, vad: Optional[torch.nn.Module] = None, seed: int = 0,
                 max_segment_frames: Optional[int] = None, use_vad_kernel: bool = True, use_decoder_kernel: bool = True,
                 asynchronous: bool = True, n_lanes: Optional[int] = None, rows_per_job: int = 32,
                 pool_rows: Optional[int] = None):
        self.S, self.C, self.packet = n_streams, n_channels, packet
        self.hga = HgaExtractorGPU(n_streams, n_channels, fs=fs)
        self.decoder = self._make_decoder(n_channels, decoder, seed)
        if vad is None:
            from .models import UnidirectionalVoiceActivityDetector
            torch.manual_seed(seed + 1)     # no trained checkpoint exists offline: seeded random weights
            vad = UnidirectionalVoiceActivityDetector(nb_layer=2, nb_hidden_units=150, nb_electrodes=n_channels)
        self.vad = vad.eval().cuda()
        # The reference's detector (2-layer LSTM + linear head, models.py:11-33) runs as ONE launch per tick on the library's own
        # kernel (dss_vad_step_dev: all streams, all frames of the packet, state carried in HBM); any other module is called as
        # given, on PyTorch-ROCm.
        from . import vad as _vad
        self.vad_gpu = _vad.make_kernel(self.vad, n_streams) if use_vad_kernel else None                    # fits() + a probe of forward()
        self.vad_state = None if self.vad_gpu else self.vad.create_new_initial_state(batch_size=n_streams, device="cuda")
        max_w = packet // max(1, int(0.01 * fs)) + 1      # frames one packet can complete (10 ms shift)
        self.gate = SpeechGateGPU(n_streams, n_channels, buffer_size, context_frames, smoothing_context, 0.6, max_frames=max_w)
        self.seg_cap = int(max_segment_frames or buffer_size)
        self.vocoder = LPCNetBatch(n_streams, 1)                  # slot = stream: vocoder state carries across segments;
        #                                                           every launch runs on a lane of it (own scratch)
        from . import decoder as _dec
        from .segment_queue import SegmentSynthesisQueue
        self._make_decoder_kernel(1, self.seg_cap, use_decoder_kernel)     # (one whole segment: the probe of forward(), and _decode())
        kernel_ok = self.dec_gpu is not None
        factory = (lambda rows, frames: _dec.BiLstmDecoderGPU(rows, frames, self.decoder)) if kernel_ok else None
        self.asynchronous = bool(asynchronous)
        self.queue = SegmentSynthesisQueue(self.gate, self.vocoder, n_channels, self.seg_cap, decoder_factory=factory,
                                           decoder_module=self.decoder, n_lanes=n_lanes if asynchronous else 1,
                                           rows_per_job=min(rows_per_job, n_streams) if asynchronous else n_streams,
                                           threaded=asynchronous, pool_rows=pool_rows)
        mean = np.zeros(n_channels) if channel_means is None else np.asarray(channel_means, dtype=np.float64)
        std = np.ones(n_channels) if channel_stds is None else np.asarray(channel_stds, dtype=np.float64)
        self.mean, self.std = torch.from_numpy(mean).cuda(), torch.from_numpy(std).cuda()
        self._zs_in_kernel = True                             # ZScoreNormalization as the epilogue of the HGA launch
        if self._zs_in_kernel:
            self.hga.set_zscore(mean, std)
        self._in = torch.empty((n_streams, packet, n_channels), dtype=torch.float64, device="cuda")
        self.frame_counter = 0
        self.last_labels = None           # raw VAD decisions and z-scored frames of the last tick (test taps)
        self.last_z = None
        self.segments_closed = 0
        self._in_wire = None

    @torch.no_grad()
    def push(self, packets: np.ndarray):
        """packets: host float64 (S, packet, C).  Returns a list of (stream, previous_frames, pcm int16 host array): the
        speech segments whose audio has FINISHED since the last call (asynchronous mode; a stream's segments always in
        closing order), or the segments that closed on this tick (asynchronous=False: the tick waits for them)."""
        self._in.copy_(torch.from_numpy(np.ascontiguousarray(packets, dtype=np.float64)))
        return self._after_extract(self.hga.extract_torch(self._in, apply_log=True))    # (S, W, C) float64

    @torch.no_grad()
    def push_wire(self, payloads: np.ndarray):
        """The same tick from the packets' bodies as they arrive on the wire: host float32 (S, C, packet), channel-major
        (``dss_amd.formats.packet_payload`` of each stream's packet, stacked).  Half the bytes cross the bus, and the per-packet
        reshape / transpose / astype(float64) of ZMQConnector.interpret_bytes (units.py:78-82) runs on the device for all
        streams at once; the frames -- and everything behind them -- are bit-identical to ``push`` on the parsed packets."""
        p = np.ascontiguousarray(payloads, dtype=np.float32)
        if p.shape != (self.S, self.C, self.packet):
            raise ValueError(f"expected payloads ({self.S}, {self.C}, {self.packet}) float32, got {p.shape}")
        if self._in_wire is None:
            self._in_wire = torch.empty((self.S, self.C, self.packet), dtype=torch.float32, device="cuda")
        self._in_wire.copy_(torch.from_numpy(p))
        return self._after_extract(self.hga.extract_wire_torch(self._in_wire, apply_log=True))

    def _after_extract(self, hga):
        W = hga.shape[1]
        if W == 0:
            return self.queue.poll()
        # post-transform (ZScoreNormalization): already applied inside the launch when the channel count allows
        z = hga if self._zs_in_kernel else ((hga - self.mean) / self.std).contiguous()
        if self.vad_gpu is not None:                                                 # units.py:433-434 in one launch
            labels = self.vad_gpu.step_torch(z)
        else:
            logits, self.vad_state = self.vad(z.to(torch.float32), self.vad_state)
            labels = torch.argmax(logits, dim=2).to(torch.int32).contiguous()
        self.last_labels, self.last_z = labels, z
        events = self.gate.push_torch(z, labels)
        self.frame_counter += W
        closers = np.flatnonzero(events[:, 0])
        if closers.size:
            streams, evs, lengths, tags = [], [], [], []
            for e in range(self.gate.E):                 # a stream closes at most one segment per tick in practice
                for s in closers:
                    if events[s, 0] > e:
                        n = int(events[s, 2 + e])
                        streams.append(int(s)); evs.append(e); lengths.append(n)
                        tags.append(self.frame_counter - n - (W - int(events[s, 1])))       # previous_frames, units.py:445
            self.segments_closed += len(streams)
            self.queue.submit(streams, evs, lengths, tags)
        return self.queue.drain() if not self.asynchronous else self.queue.poll()

    def poll(self):
        """Segments finished since the last push()/poll() (a host between ticks may call this as often as it likes)."""
        return self.queue.poll()

    def close(self):
        """Stop the queue's worker thread and release its lanes (also done when the pipeline is dropped)."""
        q = getattr(self, "queue", None)
        if q is not None:
            q.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def flush(self):
        """Wait for every segment that has closed so far; returns their (stream, previous_frames, pcm)."""
        return self.queue.drain()
