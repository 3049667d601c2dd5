"""Drop-in for the reference's Cython module ``LPCNet`` (extensions/lpcnet/LPCNet.pyx), backed by
libdss_hip.so on MI355X.

Same names, argument meaning and error behaviour as the reference wrapper:
  * ``LPCNet()``                       LPCNet.pyx:14-17  -> lpcnet_create(); MemoryError when it returns NULL
  * ``LPCNet.synthesize(features)``    LPCNet.pyx:30-40  -> float32[>=20] in, fresh int16[160] out
  * ``LPCNet.reset_decoder()``         LPCNet.pyx:23-28  -> lpcnet_init()
  * ``LPCFeatureFile``                 LPCNet.pyx:90-115 -> iterator over a raw .f32 feature file
``decode_online.py`` / ``local/units.py`` import this module by name (units.py:7,524), so putting this
directory on PYTHONPATH is all a user of the reference has to do.  The calls go through the xiph C symbols
exported by the library (cLPCNet.pxd:10-13); INTEGRATION.md shows how to compile the reference's own
LPCNet.pyx against them instead of using this ctypes stub.
"""
import numpy as np

from dss_amd import _lib as _dss
from dss_amd import lpcnet as _host


class LPCNet:
    LPCNET_FRAME_SIZE: int = 160

    def __init__(self):
        L = _dss.load()
        try:
            _host.ensure_model()        # $DSS_LPCNET_WEIGHTS or an explicit load_model(); never random weights by default
            self._st = L.lpcnet_create()
        except _dss.DssError as e:
            raise MemoryError(str(e)) from None                 # LPCNet.pyx:16-17: a failed create is a MemoryError
        if not self._st:
            raise MemoryError(L.dss_last_error().decode())      # LPCNet.pyx:16-17
        self._L = L

    def __del__(self):
        st, self._st = getattr(self, "_st", None), None
        if st:
            self._L.lpcnet_destroy(st)                          # LPCNet.pyx:19-21

    def reset_decoder(self):
        self._L.lpcnet_init(self._st)

    def synthesize(self, features):
        # the Cython signature is np.ndarray[np.float32_t, ndim=1]; mirror its buffer checks
        if not isinstance(features, np.ndarray):
            raise TypeError("Argument 'features' has incorrect type (expected numpy.ndarray, got %s)"
                            % type(features).__name__)
        if features.ndim != 1:
            raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % features.ndim)
        if features.dtype != np.float32:
            raise ValueError("Buffer dtype mismatch, expected 'float32_t' but got '%s'" % features.dtype.name)
        if features.shape[0] < 20:
            raise ValueError("features must hold at least 20 values (18 cepstra, pitch period, pitch correlation)")
        feats = np.ascontiguousarray(features)
        result = np.ones(self.LPCNET_FRAME_SIZE, dtype=np.int16, order="C")
        self._L.lpcnet_synthesize(self._st, feats.ctypes.data, result.ctypes.data, self.LPCNET_FRAME_SIZE)
        return result


class LPCFeatureEncoder:
    """Feature *encoder* (lpcnet_compute_single_frame_features, cLPCNet.pxd:15-19).  The reference uses it
    only offline in prepare_corpus.py:72-73; it is not on the synthesis path this library accelerates."""
    NB_FEATURES: int = 20
    NB_TOTAL_FEATURES: int = 36
    LPCNET_FRAME_SIZE: int = 160

    def __init__(self):
        # LPCNet.pyx:53-56: lpcnet_encoder_create() == NULL -> MemoryError.  libdss_hip.so exports the encoder symbols
        # (cLPCNet.pxd:15-19) and its create returns NULL: corpus preparation is outside the accelerated path.
        st = _dss.load().lpcnet_encoder_create()
        if not st:
            raise MemoryError(_dss.load().dss_last_error().decode())
        raise MemoryError("unexpected encoder state from libdss_hip")


class LPCFeatureFile:
    """Iterate the frames of a raw float32 feature file written by ``lpcnet_demo -features`` (36 floats per
    frame, the first 20 are what the decoder consumes)."""

    def __init__(self, filename, loop=False, nb_total_features=36):
        self.features = np.fromfile(filename, dtype=np.float32).reshape((-1, nb_total_features))
        self.index = 0
        self.loop = loop

    def __iter__(self):
        return self

    def __next__(self):
        if self.index >= len(self.features):
            raise StopIteration
        row = self.features[self.index]
        self.index += 1
        if self.loop and self.index == len(self.features):
            self.index = 0
        return row[0:20]
