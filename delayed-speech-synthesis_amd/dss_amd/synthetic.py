"""Seeded synthetic inputs of the shapes BASELINE.json / SURVEY.md section 8(d) name.

No clinical data or trained weights exist offline, so every test, golden fixture and bench line uses
these generators (numpy ``default_rng``; the same numpy build runs here and on the GPU box).
"""
from __future__ import annotations

import numpy as np

from .lpcnet_weights import synthetic_features  # noqa: F401  (config 1/2/4 input)


def synthetic_ecog(seed: int, n_samples: int = 1040, n_channels: int = 64, fs: int = 1000) -> np.ndarray:
    """Config 3 input: float64 (n_samples, n_channels) ~ N(0, 50^2) uV-scale noise plus 60/120 Hz line
    components (amplitude 20) that exercise the 118-122 Hz band-stop."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64)[:, None] / fs
    phase = rng.uniform(0, 2 * np.pi, size=(2, n_channels))
    x = rng.standard_normal((n_samples, n_channels)) * 50.0
    x += 20.0 * np.sin(2 * np.pi * 60.0 * t + phase[0])
    x += 20.0 * np.sin(2 * np.pi * 120.0 * t + phase[1])
    return np.ascontiguousarray(x)
