"""Replay a recorded session of the reference's online graph through the GPU path and diff it against the session's own
logs.  decode_online.py's loggers leave, in its run directory (decode_online.py:134-146):

    log.raw.f64   amplifier packets as received, T x C_raw float64      (BinaryLogger on ZMQConnector's output)
    log.hga.f64   z-scored high-gamma frames, frames x 64 float64       (... on HighGammaActivity's output)
    log.lpc.f32   decoded vocoder features, frames x 20 float32         (... on RecurrentNeuralDecodingModel's output)

    python -m dss_amd.replay <run dir> [--raw-columns 129] [--packet 40] [--normalization stats.npy] [--wav out.wav]

  * raw -> GPU front end (reorder + per-grid CAR + select, dss_amd.electrodes.reference_frontend) -> IIR x2 -> frames ->
    z-score, packet by packet exactly as the amplifier delivered them; compared with log.hga.f64 (bit-exact expected: the
    host-buffer entry point applies the host libm log);
  * log.lpc.f32 -> LPCNet (one decoder state for the whole session, as units.py:524) -> 16 kHz PCM, written as a wav.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import formats
from .electrodes import reference_frontend


def replay_hga(raw: np.ndarray, packet: int = 40, means=None, stds=None, fs: int = 1000) -> np.ndarray:
    """raw (T, C_raw) float64 -> z-scored frames (W, 64), packet by packet through HgaExtractorGPU."""
    from .hga import HgaExtractorGPU
    src, grid_of, comp = reference_frontend()
    ex = HgaExtractorGPU(1, len(src), fs=fs)
    ex.set_frontend(raw.shape[1], src, grid_of, comp)
    out = [ex.extract_raw(p[None])[0] for p in formats.iter_packets(raw, packet)]
    frames = np.concatenate(out) if out else np.zeros((0, len(src)))
    if means is not None:
        frames = (frames - means) / stds                       # ZScoreNormalization (common.py:375-376)
    return frames


def replay_vocoder(lpc: np.ndarray) -> np.ndarray:
    """(frames, 20) float32 -> int16 PCM, one decoder state throughout."""
    from .lpcnet import LPCNetBatch
    step = 2000
    dec = LPCNetBatch(1, step)
    pcm = [dec.synthesize(np.ascontiguousarray(lpc[None, a:a + step], dtype=np.float32))[0] for a in range(0, len(lpc), step)]
    return np.concatenate(pcm) if pcm else np.zeros(0, np.int16)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("run_dir")
    ap.add_argument("--raw-columns", type=int, default=129)
    ap.add_argument("--packet", type=int, default=40)
    ap.add_argument("--normalization", help=".npy with rows [means, stds] (baseline_offline.py:52-60)")
    ap.add_argument("--wav")
    a = ap.parse_args(argv)
    rc = 0
    raw_p, hga_p, lpc_p = (os.path.join(a.run_dir, n) for n in ("log.raw.f64", "log.hga.f64", "log.lpc.f32"))
    if os.path.exists(raw_p):
        raw = formats.read_stream_log(raw_p, a.raw_columns, np.float64)
        means = stds = None
        if a.normalization:
            st = np.load(a.normalization)
            means, stds = st[0], st[1]
        frames = replay_hga(raw, a.packet, means, stds)
        print(f"log.raw.f64: {raw.shape[0]} samples x {raw.shape[1]} columns -> {frames.shape[0]} frames")
        if os.path.exists(hga_p):
            want = formats.read_stream_log(hga_p, frames.shape[1], np.float64)
            n = min(len(want), len(frames))
            same = np.array_equal(frames[:n], want[:n])
            worst = float(np.max(np.abs(frames[:n] - want[:n]))) if n else 0.0
            print(f"log.hga.f64: {len(want)} frames logged, {n} compared: {'bit-identical' if same else 'DIFFERENT'} (max |diff| {worst:.3g})")
            rc |= 0 if same else 1
    if os.path.exists(lpc_p):
        lpc = formats.read_stream_log(lpc_p, 20, np.float32)
        pcm = replay_vocoder(lpc)
        print(f"log.lpc.f32: {len(lpc)} frames -> {len(pcm)} samples")
        if a.wav:
            from scipy.io.wavfile import write as wavwrite
            wavwrite(a.wav, 16000, pcm)
    return rc


if __name__ == "__main__":
    sys.exit(main())
