"""Device-resident ECoG -> audio pipelines built from the hot-path operators (BASELINE.json configs 3 and 5).

``SegmentPipeline``   config 3: a batch of 64-channel ECoG segments -> HGA frames -> z-score -> BiLSTM decoder
                      (PyTorch-ROCm, reference local/models.py) -> LPCNet -> 16 kHz PCM.  Nothing leaves HBM
                      between the stages.
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

from .hga import HgaExtractorGPU
from .lpcnet import FRAME_SIZE, LPCNetBatch


class _DecoderMixin:
    def _make_decoder(self, n_channels, decoder, seed):
        if decoder is None:
            from local.models import BidirectionalSpeechSynthesisModel
            torch.manual_seed(seed)         # no trained checkpoint exists offline: seeded random weights
            decoder = BidirectionalSpeechSynthesisModel(nb_layer=2, nb_hidden_units=100, nb_electrodes=n_channels)
        return decoder.eval().cuda()


class SegmentPipeline(_DecoderMixin):
    def __init__(self, batch: int, n_samples: int = 1040, n_channels: int = 64, fs: int = 1000,
                 channel_means: Optional[np.ndarray] = None, channel_stds: Optional[np.ndarray] = None,
                 decoder: Optional[torch.nn.Module] = None, seed: int = 0):
        self.B, self.n, self.C = batch, n_samples, n_channels
        self.hga = HgaExtractorGPU(batch, n_channels, fs=fs)
        self.frames = self.hga.frames_for(n_samples)
        self.decoder = self._make_decoder(n_channels, decoder, seed)
        self.vocoder = LPCNetBatch(batch, self.frames)
        mean = np.zeros(n_channels) if channel_means is None else np.asarray(channel_means, dtype=np.float64)
        std = np.ones(n_channels) if channel_stds is None else np.asarray(channel_stds, dtype=np.float64)
        self.mean = torch.from_numpy(mean).cuda()
        self.std = torch.from_numpy(std).cuda()

    @torch.no_grad()
    def __call__(self, ecog: torch.Tensor, return_intermediates: bool = False):
        """ecog: CUDA float64 (B, n_samples, C).  Returns int16 CUDA (B, frames*160)."""
        self.hga.reset()                                   # a fresh extractor per segment (prepare_corpus.py:147-176)
        self.vocoder.reset_async()                         # a fresh decoder per segment (training.py:193)
        hga = self.hga.extract_torch(ecog, apply_log=True)                    # (B, W, C) float64
        z = ((hga - self.mean) / self.std).to(torch.float32)                  # ZScoreNormalization, then .float()
        feats, _ = self.decoder(z, self.decoder.create_new_initial_state(batch_size=self.B, device="cuda"))
        pcm = self.vocoder.synthesize_torch(feats.contiguous())
        return (pcm, hga, feats) if return_intermediates else pcm


class StreamingPipeline(_DecoderMixin):
    def __init__(self, n_streams: int, n_channels: int = 64, fs: int = 1000, packet: int = 40,
                 decoder: Optional[torch.nn.Module] = None, seed: int = 0):
        self.S, self.C, self.packet = n_streams, n_channels, packet
        self.hga = HgaExtractorGPU(n_streams, n_channels, fs=fs)
        self.decoder = self._make_decoder(n_channels, decoder, seed)
        self.vocoder = LPCNetBatch(n_streams, 8)
        self._in = torch.empty((n_streams, packet, n_channels), dtype=torch.float64, device="cuda")

    @torch.no_grad()
    def push(self, packets: np.ndarray) -> np.ndarray:
        """packets: host float64 (S, packet, C), one amplifier packet per stream.  Returns host int16 (S, W*160)
        for the W frames this packet completed (1 for the very first packet, then 4)."""
        # plain pageable copies: CPU access to pinned (fine-grained) host memory is far slower than the copy itself
        self._in.copy_(torch.from_numpy(np.ascontiguousarray(packets, dtype=np.float64)))
        W = self.hga.frames_for(self.packet)
        hga = self.hga.extract_torch(self._in, apply_log=True)
        feats, _ = self.decoder(hga.to(torch.float32), self.decoder.create_new_initial_state(batch_size=self.S, device="cuda"))
        pcm = self.vocoder.synthesize_torch(feats.contiguous())
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
