"""ezmsg import guard.  The reference's units are ``ez.Unit`` subclasses (reference local/units.py:13); ezmsg is not part
of this image, so when it is missing a minimal stand-in with the same surface (Settings/State dataclasses, stream
markers, subscriber/publisher decorators, apply_settings) lets the unit classes be constructed and their async
handlers be driven directly (tests do exactly that).  With ezmsg installed the real package is used."""
from __future__ import annotations

try:  # pragma: no cover - depends on the environment
    import ezmsg.core as ez                      # type: ignore
    from ezmsg.eeg.eegmessage import TimeSeriesMessage  # type: ignore
    HAVE_EZMSG = True
except Exception:  # ImportError or partial installs
    HAVE_EZMSG = False
    import dataclasses
    from typing import Any, Optional

    class _AutoDataclass:
        def __init_subclass__(cls, **kw):
            super().__init_subclass__(**kw)
            dataclasses.dataclass(cls)

    class _Ez:
        class Settings(_AutoDataclass):
            pass

        class State:
            def __init__(self, **kw):
                for k, v in kw.items():
                    setattr(self, k, v)

        class Message(_AutoDataclass):
            pass

        class InputStream:
            def __init__(self, msg_type=None):
                self.msg_type = msg_type

        class OutputStream(InputStream):
            pass

        @staticmethod
        def subscriber(stream):
            def deco(fn):
                fn.__ez_subscribes__ = stream
                return fn
            return deco

        @staticmethod
        def publisher(stream):
            def deco(fn):
                fn.__ez_publishes__ = stream
                return fn
            return deco

        class Unit:
            SETTINGS: Any = None
            STATE: Any = None

            def __init__(self, settings=None):
                self.SETTINGS = settings
                ann = {}
                for klass in reversed(type(self).__mro__):
                    ann.update(getattr(klass, "__annotations__", {}))
                state_cls = ann.get("STATE")
                if isinstance(state_cls, str):      # `from __future__ import annotations` in the unit's module
                    import sys
                    state_cls = getattr(sys.modules.get(type(self).__module__), state_cls, None)
                self.STATE = state_cls() if isinstance(state_cls, type) else _Ez.State()

            def apply_settings(self, settings):
                self.SETTINGS = settings

            def initialize(self) -> None:
                pass

            def shutdown(self) -> None:
                pass

    ez = _Ez()

    @dataclasses.dataclass
    class TimeSeriesMessage:
        """Subset of ezmsg.eeg's TimeSeriesMessage used by the reference: data (time x channels), fs, time_dim."""
        data: Any = None
        fs: float = 1.0
        time_dim: int = 0
