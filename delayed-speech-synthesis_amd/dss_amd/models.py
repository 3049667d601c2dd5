"""Recurrent models of the online path, on PyTorch-ROCm (MIOpen LSTM).  north_star keeps the small decoder on
stock PyTorch; these classes exist so that checkpoints trained with the reference load unchanged: same class
names, constructor arguments, ``state_dict`` keys (``lstm.*``, ``classifier.*`` / ``regressor.*``) and
``create_new_initial_state`` / ``forward(x, state)`` contracts as the reference's local/models.py:11-58."""
from typing import Optional, Tuple

import torch
import torch.nn as nn

LstmState = Tuple[torch.Tensor, torch.Tensor]


class _RecurrentBase(nn.Module):
    _directions = 1

    def __init__(self, nb_layer, nb_hidden_units, nb_electrodes, dropout, bidirectional):
        super().__init__()
        self.nb_hidden_units = nb_hidden_units
        self.nb_layer = nb_layer
        self.lstm = nn.LSTM(input_size=nb_electrodes, hidden_size=nb_hidden_units, num_layers=nb_layer,
                            dropout=dropout, batch_first=True, bidirectional=bidirectional)

    def create_new_initial_state(self, batch_size: int, device: str = "cpu", req_grad: bool = False) -> LstmState:
        shape = (self._directions * self.nb_layer, batch_size, self.nb_hidden_units)
        return (torch.zeros(shape, requires_grad=req_grad, device=device),
                torch.zeros(shape, requires_grad=req_grad, device=device))

    def _run(self, head, x, state):
        if state is None:
            state = self.create_new_initial_state(batch_size=x.size(0), device=next(self.parameters()).device)
        y, new_state = self.lstm(x, state)
        return head(y), new_state


class UnidirectionalVoiceActivityDetector(_RecurrentBase):
    """2-class voice activity detector over high-gamma frames (reference local/models.py:11-33)."""

    def __init__(self, nb_layer: int = 2, nb_hidden_units: int = 512, nb_electrodes: int = 128, dropout: float = 0.0):
        super().__init__(nb_layer, nb_hidden_units, nb_electrodes, dropout, bidirectional=False)
        self.classifier = nn.Linear(in_features=nb_hidden_units, out_features=2)

    def forward(self, x: torch.Tensor, state: Optional[LstmState] = None):
        return self._run(self.classifier, x, state)


class BidirectionalSpeechSynthesisModel(_RecurrentBase):
    """High-gamma frames -> 20 LPCNet features per frame (reference local/models.py:36-58)."""
    _directions = 2

    def __init__(self, nb_layer: int = 2, nb_hidden_units: int = 100, nb_electrodes: int = 128, dropout: float = 0.0):
        super().__init__(nb_layer, nb_hidden_units, nb_electrodes, dropout, bidirectional=True)
        self.regressor = nn.Linear(in_features=2 * nb_hidden_units, out_features=20)

    def forward(self, x: torch.Tensor, state: Optional[LstmState] = None):
        return self._run(self.regressor, x, state)
