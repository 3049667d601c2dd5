"""Drop-in for the reference's Cython module ``hga_optimized`` (extensions/hga/hga_optimized.pyx).

  * ``compute_log_power_features(data, sr, window_length, window_shift)``  pyx:27-47 -- windowed mean power
    on the GPU (libdss_hip.so), float64, bit-identical output.
  * ``WarmStartFrameBuffer(frame_length, frame_shift, fs, nb_channels)``   pyx:50-131 -- the host-side frame
    assembler with the reference's three cases and its aliasing behaviour (the remainder is a view of the last
    returned array; in CASE 1 the input itself is returned).  It only moves rows around, so it stays on the
    host for callers that use the two functions separately; the fused GPU extractor
    (dss_amd.hga.HgaExtractorGPU) keeps the same buffer on the device.
"""
import numpy as np

from dss_amd import hga as _hga


def compute_log_power_features(data, sr, window_length, window_shift):
    data = np.asarray(data)
    if data.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'double' but got '%s'" % data.dtype.name)
    # Cython receives window_length / window_shift as C floats
    return _hga.log_power(data, int(sr), float(np.float32(window_length)), float(np.float32(window_shift)))


class WarmStartFrameBuffer:
    def __init__(self, frame_length, frame_shift, fs, nb_channels):
        fl32, fs32 = np.float32(frame_length), np.float32(frame_shift)
        shift = int(np.float32(fs32 * np.float32(fs)))                       # <int>(frame_shift * fs), pyx:72
        self.frame_length_in_samples = int(np.float32(fl32 * np.float32(fs)))   # pyx:73
        self.overlap = self.frame_length_in_samples - shift
        self.nb_channels = int(nb_channels)
        self.first_frame = True
        self.remainder_data = np.zeros((self.overlap, self.nb_channels), dtype=np.float64)

    def reset(self):
        self.first_frame = True
        self.remainder_data[:, :] = 0                                        # in place, like pyx:92-94

    def insert(self, data):
        data = np.asarray(data)
        if data.dtype != np.float64 or data.ndim != 2:
            raise ValueError("Buffer dtype mismatch, expected 'double' 2-D array")
        n = data.shape[0]
        if self.first_frame and n >= self.frame_length_in_samples:           # CASE 1
            self.first_frame = False
            self.remainder_data = data[-self.overlap:, :]
            return data
        if self.first_frame:                                                 # CASE 2: left zero padding
            out = np.empty((self.frame_length_in_samples, data.shape[1]), dtype=np.float64)
            pad = self.frame_length_in_samples - n
            out[:pad] = 0
            out[pad:] = data
            self.first_frame = False
        else:                                                                # CASE 3: prepend remainder
            out = np.empty((self.overlap + n, data.shape[1]), dtype=np.float64)
            out[:self.overlap] = self.remainder_data
            out[self.overlap:] = data
        self.remainder_data = out[-self.overlap:, :]
        return out
