"""ctypes bindings of oracle/liboracle.so for tests / smoke / cpu_baseline (checker only)."""
import ctypes as C

import numpy as np

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, so_path):
        L = self.lib = C.CDLL(so_path)
        L.oracle_sosfilt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_framebuf_create.restype = C.c_void_p
        L.oracle_framebuf_create.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int]
        L.oracle_framebuf_destroy.argtypes = [C.c_void_p]
        L.oracle_framebuf_reset.argtypes = [C.c_void_p]
        L.oracle_framebuf_insert.restype = C.c_int
        L.oracle_framebuf_insert.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_num_windows.restype = C.c_int
        L.oracle_num_windows.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
        for f in (L.oracle_log_power, L.oracle_mean_power):
            f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.oracle_hga_extract.restype = C.c_int
        L.oracle_hga_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.oracle_lpcnet_model_load.restype = C.c_void_p
        L.oracle_lpcnet_model_load.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_lpcnet_model_free.argtypes = [C.c_void_p]
        L.oracle_lpcnet_table.restype = C.POINTER(C.c_float)
        L.oracle_lpcnet_table.argtypes = [C.c_void_p, C.c_int]
        L.oracle_lpcnet_create.restype = C.c_void_p
        L.oracle_lpcnet_create.argtypes = [C.c_void_p]
        L.oracle_lpcnet_destroy.argtypes = [C.c_void_p]
        L.oracle_lpcnet_init.argtypes = [C.c_void_p]
        L.oracle_lpcnet_set_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.oracle_lpcnet_synthesize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_lpcnet_frame_network.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_lpcnet_synthesize_utterance.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_lpcnet_tap.restype = C.POINTER(C.c_float)
        L.oracle_lpcnet_tap.argtypes = [C.c_void_p, C.c_int]
        L.oracle_lpc_from_cepstrum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_lin2ulaw.restype = C.c_int
        L.oracle_lin2ulaw.argtypes = [C.c_float]
        L.oracle_ulaw2lin.restype = C.c_float
        L.oracle_ulaw2lin.argtypes = [C.c_float]
        for f in (L.oracle_tanh_approx, L.oracle_sigmoid_approx):
            f.restype = C.c_float
            f.argtypes = [C.c_void_p, C.c_float]
        L.oracle_kiss99_seed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_kiss99_srand.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.oracle_kiss99_draw.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        L.oracle_celt_lpc.restype = C.c_float
        L.oracle_celt_lpc.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_lpcnet_set_forced.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        L.oracle_lpcnet_sample_step.restype = C.c_int
        L.oracle_lpcnet_sample_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_lpcnet_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

    # ---- HGA ---------------------------------------------------------------------------------
    def sosfilt(self, sos, x, zi):
        """In-place like scipy.signal.sosfilt(sos, x, axis=0, zi=zi); returns (y, zi_out)."""
        x = np.array(x, dtype=np.float64, order="C")
        zi = np.array(zi, dtype=np.float64, order="C")
        sos = np.ascontiguousarray(sos, dtype=np.float64)
        self.lib.oracle_sosfilt(_p(sos), sos.shape[0], _p(x), x.shape[0], x.shape[1], _p(zi))
        return x, zi

    def log_power(self, data, sr=1000, wl=0.05, ws=0.01, mean_only=False):
        data = np.ascontiguousarray(data, dtype=np.float64)
        W = self.lib.oracle_num_windows(data.shape[0], sr, wl, ws)
        out = np.empty((max(W, 0), data.shape[1]), dtype=np.float64)
        f = self.lib.oracle_mean_power if mean_only else self.lib.oracle_log_power
        f(_p(data), data.shape[0], data.shape[1], sr, wl, ws, _p(out))
        return out

    class FrameBuffer:
        def __init__(self, lib, wl, ws, fs, C_):
            self.lib, self.C, self.fs, self.wl, self.ws = lib, C_, fs, wl, ws
            self.h = lib.oracle_framebuf_create(wl, ws, fs, C_)

        def insert(self, data):
            data = np.ascontiguousarray(data, dtype=np.float64)
            out = np.empty((data.shape[0] + 2 * int(self.wl * self.fs) + 8, self.C), dtype=np.float64)
            rows = self.lib.oracle_framebuf_insert(self.h, _p(data), data.shape[0], _p(out))
            return out[:rows].copy()

        def reset(self):
            self.lib.oracle_framebuf_reset(self.h)

        def __del__(self):
            self.lib.oracle_framebuf_destroy(self.h)

    def framebuffer(self, wl=0.05, ws=0.01, fs=1000, C_=64):
        return Oracle.FrameBuffer(self.lib, wl, ws, fs, C_)

    class Extractor:
        """oracle counterpart of HighGammaExtractor.extract_features (no pre/post transforms)."""

        def __init__(self, orc, sos_hg, sos_fh, zi_hg, zi_fh, C_, fs=1000, wl=0.05, ws=0.01):
            self.o, self.C, self.fs, self.wl, self.ws = orc, C_, fs, wl, ws
            self.sos_hg = np.ascontiguousarray(sos_hg, dtype=np.float64)
            self.sos_fh = np.ascontiguousarray(sos_fh, dtype=np.float64)
            self.zi_hg = np.ascontiguousarray(np.repeat(zi_hg[:, :, None], C_, axis=2))
            self.zi_fh = np.ascontiguousarray(np.repeat(zi_fh[:, :, None], C_, axis=2))
            self.fb = orc.framebuffer(wl, ws, fs, C_)

        def extract(self, data):
            data = np.array(data, dtype=np.float64, order="C")
            n = data.shape[0]
            out = np.empty((n // 1 + 8, self.C), dtype=np.float64)
            W = self.o.lib.oracle_hga_extract(_p(self.sos_hg), _p(self.sos_fh), self.sos_hg.shape[0],
                                              _p(self.zi_hg), _p(self.zi_fh), self.fb.h, _p(data), n, self.C,
                                              self.fs, self.wl, self.ws, _p(out))
            return out[:W].copy()

    def extractor(self, filt, C_, **kw):
        return Oracle.Extractor(self, filt["sos_hg"], filt["sos_fh"], filt["zi_hg"], filt["zi_fh"], C_, **kw)

    # ---- LPCNet ------------------------------------------------------------------------------
    def lpcnet_model(self, blob: bytes):
        m = self.lib.oracle_lpcnet_model_load(blob, len(blob))
        if not m:
            raise ValueError("oracle rejected the blob")
        return m

    def lpcnet_table(self, model, which, n):
        return np.ctypeslib.as_array(self.lib.oracle_lpcnet_table(model, which), shape=(n,)).copy()

    def lpcnet_utterance(self, model, features):
        features = np.ascontiguousarray(features, dtype=np.float32)
        nf = features.shape[0]
        pcm = np.zeros(nf * 160, dtype=np.int16)
        self.lib.oracle_lpcnet_synthesize_utterance(model, _p(features), nf, features.shape[1], _p(pcm))
        return pcm

    class Decoder:
        def __init__(self, lib, model, trace_cap=0):
            self.lib = lib
            self.h = lib.oracle_lpcnet_create(model)
            self.trace_exc = self.trace_pcm = None
            if trace_cap:
                self.trace_exc = np.zeros(trace_cap, np.uint8)
                self.trace_pcm = np.zeros(trace_cap, np.float32)
                lib.oracle_lpcnet_set_trace(self.h, _p(self.trace_exc), _p(self.trace_pcm), trace_cap)

        def force(self, exc, want_logits=True):
            """Teacher forcing: sample k takes exc[k]; all 255 node logits of each forced sample are recorded."""
            self.forced_exc = np.ascontiguousarray(exc, dtype=np.uint8)
            self.forced_logits = np.zeros((len(self.forced_exc), 256), np.float32) if want_logits else None
            self.lib.oracle_lpcnet_set_forced(self.h, _p(self.forced_exc),
                                              _p(self.forced_logits) if want_logits else None, len(self.forced_exc))

        def set_state(self, gru_a=None, gru_b=None, cond_a=None, cond_b=None):
            arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (gru_a, gru_b, cond_a, cond_b)]
            self.lib.oracle_lpcnet_set_state(self.h, *[None if a is None else _p(a) for a in arrs])

        def sample_step(self, last_exc, sig_ulaw, pred_ulaw):
            """One sample-network step on the current state: all 255 node logits ([0] unused)."""
            logits = np.zeros(256, np.float32)
            self.lib.oracle_lpcnet_sample_step(self.h, int(last_exc), int(sig_ulaw), int(pred_ulaw), _p(logits))
            return logits

        def synthesize(self, feat):
            feat = np.ascontiguousarray(feat, dtype=np.float32)
            out = np.ones(160, dtype=np.int16)
            self.lib.oracle_lpcnet_synthesize(self.h, _p(feat), _p(out), 160)
            return out

        def frame_network(self, feat):
            feat = np.ascontiguousarray(feat, dtype=np.float32)
            self.lib.oracle_lpcnet_frame_network(self.h, _p(feat))

        def tap(self, which, n):
            return np.ctypeslib.as_array(self.lib.oracle_lpcnet_tap(self.h, which), shape=(n,)).copy()

        def reset(self):
            self.lib.oracle_lpcnet_init(self.h)

        def __del__(self):
            self.lib.oracle_lpcnet_destroy(self.h)

    def decoder(self, model, trace_cap=0):
        return Oracle.Decoder(self.lib, model, trace_cap)

    # ---- known-answer hooks -------------------------------------------------------------------
    def kiss99(self, seed4=None, srand: bytes = None):
        ctx = np.zeros(4, np.uint32)
        if srand is not None:
            self.lib.oracle_kiss99_srand(_p(ctx), srand, len(srand))
        else:
            self.lib.oracle_kiss99_seed(_p(ctx), *[int(v) for v in seed4])
        return ctx

    def kiss99_draw(self, ctx, n, keep=1):
        out = np.zeros(keep, np.uint32)
        self.lib.oracle_kiss99_draw(_p(ctx), n, _p(out), keep)
        return out

    def celt_lpc(self, ac, p=16):
        ac = np.ascontiguousarray(ac, dtype=np.float32)
        lpc = np.zeros(p, np.float32)
        err = self.lib.oracle_celt_lpc(_p(lpc), _p(ac), p)
        return lpc, float(err)

    def lpc_from_cepstrum(self, model, cep):
        cep = np.ascontiguousarray(cep, dtype=np.float32)
        lpc = np.zeros(16, np.float32)
        self.lib.oracle_lpc_from_cepstrum(model, _p(lpc), _p(cep))
        return lpc
