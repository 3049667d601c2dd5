"""High-gamma feature extraction on MI355X: Python host side of Part 3 of include/dss_hip.h."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib


def design_filters(fs: int = 1000, l_freq: float = 70, h_freq: float = 170, order: int = 8):
    """The two SOS cascades HighGammaExtractor builds (local/units.py:124-126): Butterworth band-pass
    l_freq..h_freq and band-stop 118..122 Hz.  The reference asks mne.filter.create_filter(method='iir',
    iir_params={'order': 8, 'ftype': 'butter'}); mne resolves that to scipy.signal.iirfilter(order, Wn,
    btype, ftype='butter', output='sos'), which is called directly here (mne is not a dependency)."""
    from scipy.signal import iirfilter, sosfilt_zi
    nyq = fs / 2.0
    hg = iirfilter(order, [l_freq / nyq, h_freq / nyq], btype="bandpass", ftype="butter", output="sos")
    fh = iirfilter(order, [118 / nyq, 122 / nyq], btype="bandstop", ftype="butter", output="sos")
    return hg, fh, sosfilt_zi(hg), sosfilt_zi(fh)


_TABLES = None
_checked = False


def reference_filters(fs: int = 1000, l_freq: float = 70, h_freq: float = 170):
    """The filters HighGammaExtractor uses.  For the reference's configuration (fs 1000, 70-170 Hz: units.py:102,
    config/debug_settings.ini:19) the 2 x 8 x 6 coefficient tables and their unit-step states SHIPPED with the package
    (dss_amd/data/, the tables the golden HGA frames were generated with) are returned, so the bit-match with the
    reference chain does not depend on the scipy build on the box; design_filters() is compared with them once and a
    difference is reported.  Any other configuration is designed with scipy."""
    global _TABLES, _checked
    if (int(fs), float(l_freq), float(h_freq)) != (1000, 70.0, 170.0):
        return design_filters(fs, l_freq, h_freq)
    if _TABLES is None:
        import os
        with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "hga_filters_fs1000_70_170.npz")) as g:
            _TABLES = tuple(np.ascontiguousarray(g[k], dtype=np.float64) for k in ("sos_hg", "sos_fh", "zi_hg", "zi_fh"))
    if not _checked:
        _checked = True
        try:
            same = all(np.array_equal(a, b) for a, b in zip(design_filters(fs, l_freq, h_freq), _TABLES))
        except Exception:           # scipy absent or failing: the tables are all that is needed
            same = True
        if not same:
            import warnings
            warnings.warn("scipy on this machine designs different Butterworth sections than the shipped tables "
                          "(scipy version skew); using the shipped tables, which the golden HGA frames were made with",
                          RuntimeWarning, stacklevel=2)
    return tuple(a.copy() for a in _TABLES)


def num_windows(T: int, sr: int, wl: float, ws: float) -> int:
    return int(_lib.load().dss_hga_num_windows(int(T), int(sr), wl, ws))


def log_power(data: np.ndarray, sr: int, wl: float, ws: float) -> np.ndarray:
    """compute_log_power_features (hga_optimized.pyx:27-47) on the GPU; float64 (T, C) -> (W, C)."""
    L = _lib.require_gpu()
    d = np.ascontiguousarray(data, dtype=np.float64)
    if d.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % d.ndim)
    W = max(num_windows(d.shape[0], sr, wl, ws), 0)
    out = np.empty((W, d.shape[1]), dtype=np.float64)
    if W:
        _lib.check(L.dss_hga_log_power(d.ctypes.data, d.shape[0], d.shape[1], int(sr), wl, ws, out.ctypes.data))
    return out


class HgaExtractorGPU:
    """n_streams independent HighGammaExtractor states (filter state + warm-start frame buffer) on one GPU."""

    def __init__(self, n_streams: int, n_channels: int, fs: int = 1000, window_length: float = 0.05,
                 window_shift: float = 0.01, filters=None):
        L = _lib.require_gpu()
        self._L = L
        hg, fh, zi_hg, zi_fh = filters if filters is not None else reference_filters(fs)
        hg, fh = (np.ascontiguousarray(a, dtype=np.float64) for a in (hg, fh))
        zi_hg, zi_fh = (np.ascontiguousarray(a, dtype=np.float64) for a in (zi_hg, zi_fh))
        if hg.shape != fh.shape or hg.shape[1] != 6:
            raise ValueError("both filters must be (n_sections, 6) SOS arrays of equal length")
        self.S, self.C, self.fs = int(n_streams), int(n_channels), int(fs)
        self.wl, self.ws = float(window_length), float(window_shift)
        self._h = L.dss_hga_create(self.S, self.C, self.fs, self.wl, self.ws, hg.shape[0], hg.ctypes.data,
                                   fh.ctypes.data, zi_hg.ctypes.data, zi_fh.ctypes.data)
        if not self._h:
            raise _lib.DssError(L.dss_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.dss_hga_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        _lib.check(self._L.dss_hga_reset(self._h))

    def frames_for(self, n: int) -> int:
        return int(self._L.dss_hga_frames_for(self._h, int(n)))

    def set_zscore(self, means=None, stds=None):
        """ZScoreNormalization (local/common.py:367-376) as the epilogue of the extractor's own launch:
        frames -> (frames - means) / stds.  ``None`` clears it.  (dss_hga_set_zscore)"""
        if means is None:
            _lib.check(self._L.dss_hga_set_zscore(self._h, None, None))
            return
        m = np.ascontiguousarray(means, dtype=np.float64)
        sd = np.ascontiguousarray(stds, dtype=np.float64)
        if m.shape != (self.C,) or sd.shape != (self.C,):
            raise ValueError(f"means / stds must have shape ({self.C},)")
        _lib.check(self._L.dss_hga_set_zscore(self._h, m.ctypes.data, sd.ctypes.data))

    def _force_path(self, path: int):
        """Tests / A-B timing only: 0 choose, 1 hga_fused_kernel, 2 three launches (dss_selftest_hga_force_path)."""
        _lib.check(self._L.dss_selftest_hga_force_path(self._h, int(path)))

    def extract(self, data: np.ndarray) -> np.ndarray:
        """(S, n, C) float64 host -> (S, W, C) float64 host, bit-identical to the reference chain."""
        d = np.ascontiguousarray(data, dtype=np.float64)
        if d.ndim == 2:
            d = d[None]
        if d.shape[0] != self.S or d.shape[2] != self.C:
            raise ValueError(f"expected ({self.S}, n, {self.C}), got {d.shape}")
        n = d.shape[1]
        W = self.frames_for(n)
        out = np.empty((self.S, max(W, 1), self.C), dtype=np.float64)
        got = _lib.check(self._L.dss_hga_extract(self._h, d.ctypes.data, n, out.ctypes.data))
        assert got == W
        return out[:, :W].reshape(self.S, W, self.C) if W else np.empty((self.S, 0, self.C))

    # ---- fused front end (column reorder + per-grid CAR + channel selection) -------------------------------
    def set_frontend(self, c_raw: int, src_col, grid_of, comp_lists):
        """output channel c = raw[src_col[c]] - mean(raw[comp_lists[grid_of[c]]]) (sequential mean, numpy order)."""
        src = np.ascontiguousarray(src_col, dtype=np.int32)
        gof = np.ascontiguousarray(grid_of, dtype=np.int32)
        assert src.shape == (self.C,) and gof.shape == (self.C,)
        comp = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.int32) for c in comp_lists])
                                    if len(comp_lists) else np.zeros(1, np.int32), dtype=np.int32)
        off = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(c) for c in comp_lists])]), dtype=np.int32)
        _lib.check(self._L.dss_hga_set_frontend(self._h, int(c_raw), src.ctypes.data, gof.ctypes.data, len(comp_lists),
                                                comp.ctypes.data, off.ctypes.data))
        self.c_raw = int(c_raw)

    def set_frontend_from_transforms(self, c_raw: int, select_all, car, select_sub) -> None:
        """Build the fused front end from the reference's three pre-transform objects (decode_online.py:65-85):
        SelectElectrodesFromBothGrids -> CommonAverageReferencing -> SelectElectrodesOverSpeechAreas."""
        from .electrodes import frontend_from_transforms
        self.set_frontend(c_raw, *frontend_from_transforms(select_all, car, select_sub))

    def extract_raw(self, raw: np.ndarray) -> np.ndarray:
        """(S, n, c_raw) raw amplifier packets -> (S, W, C) frames, front end + filters + log power on the GPU."""
        d = np.ascontiguousarray(raw, dtype=np.float64)
        if d.ndim == 2:
            d = d[None]
        if d.shape[0] != self.S or d.shape[2] != self.c_raw:
            raise ValueError(f"expected ({self.S}, n, {self.c_raw}), got {d.shape}")
        n = d.shape[1]
        W = self.frames_for(n)
        out = np.empty((self.S, max(W, 1), self.C), dtype=np.float64)
        got = _lib.check(self._L.dss_hga_extract_raw(self._h, d.ctypes.data, n, out.ctypes.data))
        assert got == W
        return out[:, :W].reshape(self.S, W, self.C) if W else np.empty((self.S, 0, self.C))

    def extract_raw_torch(self, raw, apply_log: bool = True, stream=None):
        import torch
        assert raw.is_cuda and raw.dtype == torch.float64 and raw.is_contiguous() and raw.shape[2] == self.c_raw
        n = raw.shape[1]
        W = self.frames_for(n)
        out = torch.empty((self.S, W, self.C), dtype=torch.float64, device=raw.device)
        s = torch.cuda.current_stream(raw.device).cuda_stream if stream is None else stream
        got = _lib.check(self._L.dss_hga_extract_raw_dev(self._h, raw.data_ptr(), n, out.data_ptr(), int(apply_log), s))
        assert got == W
        return out

    def extract_wire_torch(self, payload, apply_log: bool = True, stream=None):
        """Payloads in wire format: CUDA float32 (S, c_in, n), channel-major -- the bodies of the amplifier's packets
        (``dss_amd.formats.packet_payload``), c_in = c_raw with a front end configured, else C -> (S, W, C) float64 frames.  The
        reshape / transpose / astype(float64) of ZMQConnector.interpret_bytes (units.py:78-82) runs on the device."""
        import torch
        c_in = getattr(self, "c_raw", None) or self.C
        assert payload.is_cuda and payload.dtype == torch.float32 and payload.is_contiguous()
        if payload.dim() != 3 or payload.shape[0] != self.S or payload.shape[1] != c_in:
            raise ValueError(f"expected ({self.S}, {c_in}, n) float32, got {tuple(payload.shape)}")
        n = payload.shape[2]
        W = self.frames_for(n)
        out = torch.empty((self.S, W, self.C), dtype=torch.float64, device=payload.device)
        s = torch.cuda.current_stream(payload.device).cuda_stream if stream is None else stream
        got = _lib.check(self._L.dss_hga_extract_wire_dev(self._h, payload.data_ptr(), n, out.data_ptr(), int(apply_log), s))
        assert got == W
        return out

    def extract_torch(self, data, apply_log: bool = True, out=None, stream=None):
        """Device-resident: (S, n, C) float64 CUDA tensor -> (S, W, C) float64 CUDA tensor."""
        import torch
        assert data.is_cuda and data.dtype == torch.float64 and data.is_contiguous()
        n = data.shape[1]
        W = self.frames_for(n)
        if out is None:
            out = torch.empty((self.S, W, self.C), dtype=torch.float64, device=data.device)
        s = torch.cuda.current_stream(data.device).cuda_stream if stream is None else stream
        got = _lib.check(self._L.dss_hga_extract_dev(self._h, data.data_ptr(), n, out.data_ptr(), int(apply_log), s))
        assert got == W
        return out
