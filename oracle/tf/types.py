"""Type aliases used in the reference's annotations (``door_models/pf.py:64,69``,
``tasks/_door.py:299``)."""
from typing import Dict, NamedTuple, Union

import numpy as np
import torch

NumpyDict = Dict[str, np.ndarray]
TorchDict = Dict[str, torch.Tensor]
NumpyArrayOrDict = Union[np.ndarray, NumpyDict]
TorchTensorOrDict = Union[torch.Tensor, TorchDict]

StatesNumpy = np.ndarray
StatesTorch = torch.Tensor
ObservationsNumpy = NumpyArrayOrDict
ObservationsTorch = TorchTensorOrDict
ControlsNumpy = NumpyArrayOrDict
ControlsTorch = TorchTensorOrDict
ScaleTrilTorch = torch.Tensor
CovarianceTorch = torch.Tensor


class TrajectoryNumpy(NamedTuple):
    """Positional 3-tuple, unpackable (``eval_helpers.py:90-95``)."""

    states: StatesNumpy
    observations: ObservationsNumpy
    controls: ControlsNumpy
