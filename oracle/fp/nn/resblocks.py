"""Two-layer residual blocks (fannypack.nn.resblocks restated).

``y = act(block2(act(block1(x))) + x)`` with ``act`` = ReLU.  Sub-module names
``block1``/``block2`` fix the ``state_dict`` key scheme.
"""
import torch
import torch.nn as nn


class _Residual(nn.Module):
    def __init__(self, block1: nn.Module, block2: nn.Module):
        super().__init__()
        self.block1 = block1
        self.block2 = block2

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = torch.relu(self.block1(x))
        return torch.relu(self.block2(h) + x)


class Linear(_Residual):
    def __init__(self, units: int, bottleneck_units: int = None):
        inner = units if bottleneck_units is None else bottleneck_units
        super().__init__(nn.Linear(units, inner), nn.Linear(inner, units))


class Conv2d(_Residual):
    def __init__(self, channels: int, bottleneck_channels: int = None, kernel_size: int = 3):
        inner = channels if bottleneck_channels is None else bottleneck_channels
        pad = kernel_size // 2
        super().__init__(
            nn.Conv2d(channels, inner, kernel_size, padding=pad),
            nn.Conv2d(inner, channels, kernel_size, padding=pad),
        )
