"""Type aliases of the torchfilter API surface the reference's models are annotated with
(``/root/reference/crossmodal/door_models/pf.py:64,69``, ``tasks/_door.py:299``)."""
from typing import Dict, NamedTuple, Union

import numpy as np
import torch

NumpyDict = Dict[str, np.ndarray]
TorchDict = Dict[str, torch.Tensor]
StatesNumpy = np.ndarray
StatesTorch = torch.Tensor
ObservationsNumpy = Union[np.ndarray, NumpyDict]
ObservationsTorch = Union[torch.Tensor, TorchDict]
ControlsNumpy = Union[np.ndarray, NumpyDict]
ControlsTorch = Union[torch.Tensor, TorchDict]
ScaleTrilTorch = torch.Tensor
CovarianceTorch = torch.Tensor


class TrajectoryNumpy(NamedTuple):
    states: StatesNumpy
    observations: ObservationsNumpy
    controls: ControlsNumpy
