"""Stand-in for the one unit of the user's ``local.units`` that dss_amd.units SUBCLASSES (``gpu_decoding_unit``).

The reference's ``local/units.py`` cannot be imported in this image (zmq, mne, ezmsg are absent), so tests need something to
apply the factory to.  This is that something and nothing more: the settings / state field names the reference's unit has
(units.py:450-470) and a plain module call per segment.  ``python -m dss_amd.run`` wraps the USER'S class and never imports
this file."""
from __future__ import annotations

from dataclasses import replace
from typing import AsyncGenerator, Callable, Optional

import numpy as np

from ._ez import TimeSeriesMessage, ez


class RecurrentNeuralDecodingModelSettings(ez.Settings):
    path_to_model_weights: Optional[str]
    model: Callable
    params: Optional[dict]
    config_filename: Optional[str] = None


class RecurrentNeuralDecodingModelState(ez.State):
    decoding_model = None
    device: Optional[str] = None
    H = None


class StandInDecodingUnit(ez.Unit):
    SETTINGS: RecurrentNeuralDecodingModelSettings
    STATE: RecurrentNeuralDecodingModelState
    INPUT = ez.InputStream(TimeSeriesMessage)
    OUTPUT = ez.OutputStream(TimeSeriesMessage)

    def initialize(self) -> None:
        import torch
        cfg, st = self.SETTINGS, self.STATE
        st.device = "cuda" if torch.cuda.is_available() else "cpu"
        net = cfg.model(**(cfg.params or {}))
        if cfg.path_to_model_weights:
            net.load_state_dict(torch.load(cfg.path_to_model_weights, map_location=st.device))
        st.decoding_model = net.to(st.device).eval()
        st.H = net.create_new_initial_state(batch_size=1, device=st.device)

    @ez.subscriber(INPUT)
    @ez.publisher(OUTPUT)
    async def decode(self, msg: TimeSeriesMessage) -> AsyncGenerator:
        import torch
        st = self.STATE
        x = torch.from_numpy(np.asarray(msg.data)[None]).float().to(st.device)
        with torch.no_grad():
            y, _ = st.decoding_model(x, st.H)
        st.H = st.decoding_model.create_new_initial_state(batch_size=1, device=st.device)
        yield self.OUTPUT, replace(msg, data=y[0].cpu().numpy(), fs=100)
