"""Wire and disk formats either side of the hot path, so recorded sessions of the reference can
be replayed through the GPU path and diffed against its logs (tools/replay_logs.py).

  amplifier packet   '=BBB HH' header (descriptor 4, supplement 1, dtype 2, n_channels, n_samples) + float32
                     [n_channels][n_samples]; ZMQ topic = the first three header bytes
                     (local/units.py:48,63,78-82; development_amplifier.py:14-25)
  raw stream logs    BinaryLogger dumps message.data.tobytes() back to back: log.raw.f64 (T x C_raw float64),
                     log.hga.f64 (frames x 64 float64), log.lpc.f32 (frames x 20 float32)
                     (local/units.py:264-270; decode_online.py:134-146)
  VAD label file     one "start<TAB>stop<TAB><n> frames" line per segment, seconds with 2 decimals (units.py:311-319)
  vocoder features   xiph .f32: 36 float32 per frame, the first 20 are consumed (LPCNet.pyx:90-115)
  audio sink         raw s16le mono 16 kHz on stdout for SoX `play` (units.py:550-552; replicate.sh:115-116)
"""
from __future__ import annotations

import struct
from typing import Iterator, List, Tuple

import numpy as np

PACKET_HEADER = struct.Struct("=BBB HH")
PACKET_TOPIC = bytes((4, 1, 2))


def parse_packet(data: bytes) -> np.ndarray:
    """One amplifier packet -> float64 (n_samples, n_channels), C order (what ZMQConnector hands downstream)."""
    if len(data) < PACKET_HEADER.size:
        raise ValueError("packet shorter than its 7-byte header")
    _, _, _, n_ch, n_smp = PACKET_HEADER.unpack_from(data)
    body = np.frombuffer(data, dtype="<f4", count=n_ch * n_smp, offset=PACKET_HEADER.size)
    return np.ascontiguousarray(body.reshape(n_ch, n_smp).T, dtype=np.float64)


def packet_payload(data: bytes) -> np.ndarray:
    """One amplifier packet -> its body as it is: float32 (n_channels, n_samples), a VIEW of the bytes (no transpose, no
    conversion).  S of them stacked are what ``HgaExtractorGPU.extract_wire_torch`` / ``GatedStreamingPipeline.push_wire`` take:
    the transpose and the float64 conversion of ``parse_packet`` then run on the device."""
    if len(data) < PACKET_HEADER.size:
        raise ValueError("packet shorter than its 7-byte header")
    _, _, _, n_ch, n_smp = PACKET_HEADER.unpack_from(data)
    return np.frombuffer(data, dtype="<f4", count=n_ch * n_smp, offset=PACKET_HEADER.size).reshape(n_ch, n_smp)


def build_packet(samples: np.ndarray) -> bytes:
    """Inverse of parse_packet: (n_samples, n_channels) -> packet bytes (development_amplifier.py:14-25)."""
    s = np.asarray(samples)
    return PACKET_HEADER.pack(4, 1, 2, s.shape[1], s.shape[0]) + np.ascontiguousarray(s.T, dtype="<f4").tobytes()


def read_stream_log(path, columns: int, dtype) -> np.ndarray:
    """log.raw.f64 / log.hga.f64 / log.lpc.f32 -> (rows, columns)."""
    a = np.fromfile(path, dtype=dtype)
    if a.size % columns:
        raise ValueError(f"{path}: {a.size} values do not divide into rows of {columns}")
    return a.reshape(-1, columns)


def append_stream_log(fh, data: np.ndarray) -> None:
    fh.write(np.ascontiguousarray(data).tobytes())


def read_vad_labels(path) -> List[Tuple[float, float, int]]:
    out = []
    with open(path) as f:
        for line in f:
            a, b, c = line.rstrip("\n").split("\t")
            out.append((float(a), float(b), int(c.split()[0])))
    return out


def format_vad_label(previous_frames: float, n_frames: int, frameshift: float = 0.01) -> str:
    return f"{previous_frames * frameshift:.02f}\t{(previous_frames + n_frames) * frameshift:.02f}\t{n_frames} frames\n"


def read_feature_file(path, nb_total_features: int = 36) -> np.ndarray:
    """xiph .f32 feature file -> (frames, 20): the columns lpcnet_synthesize consumes."""
    return read_stream_log(path, nb_total_features, "<f4")[:, :20]


def pcm_to_s16le(pcm: np.ndarray) -> bytes:
    return np.ascontiguousarray(pcm, dtype="<i2").tobytes()


def iter_packets(raw_log: np.ndarray, n_samples: int = 40) -> Iterator[np.ndarray]:
    """Cut a log.raw.f64 array back into the packets the amplifier delivered (debug_settings.ini: 40 samples)."""
    for a in range(0, raw_log.shape[0] - n_samples + 1, n_samples):
        yield raw_log[a:a + n_samples]
