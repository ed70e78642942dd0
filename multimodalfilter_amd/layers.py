"""Network building blocks with the reference's ``state_dict`` key scheme.

Host-side mirror of ``/root/reference/crossmodal/{door,push}_models/layers.py`` and the
``fannypack.nn.resblocks`` blocks they are made of.  These modules
*own the parameters*; the per-particle arithmetic runs in the HIP kernels
(``csrc/particle_net.hip``) from a fragment-ordered copy of them (``engine.PackedParticleNet``).
Per-trajectory encoders (N rows, not N*M) are evaluated with these modules directly.
"""
import torch
import torch.nn as nn


class ResLinear(nn.Module):
    """``relu(block2(relu(block1(x))) + x)`` -- fannypack ``resblocks.Linear``."""

    def __init__(self, units: int):
        super().__init__()
        self.block1 = nn.Linear(units, units)
        self.block2 = nn.Linear(units, units)

    def forward(self, x):
        return torch.relu(self.block2(torch.relu(self.block1(x))) + x)


class ResConv2d(nn.Module):
    """fannypack ``resblocks.Conv2d`` (same-padding, two convolutions)."""

    def __init__(self, channels: int, kernel_size: int = 3):
        super().__init__()
        self.block1 = nn.Conv2d(channels, channels, kernel_size, padding=kernel_size // 2)
        self.block2 = nn.Conv2d(channels, channels, kernel_size, padding=kernel_size // 2)

    def forward(self, x):
        return torch.relu(self.block2(torch.relu(self.block1(x))) + x)


def vector_encoder(in_dim: int, units: int = 64) -> nn.Sequential:
    """``state_layers`` / ``control_layers`` / ``observation_{pos,sensors}_layers``
    (``door_models/layers.py:11-40,66-95``): Linear, ReLU, ResLinear."""
    return nn.Sequential(nn.Linear(in_dim, units), nn.ReLU(), ResLinear(units))


class DualSpanningAvgPool(nn.Module):
    """``push_models/layers.py:43-65``: full-height and full-width average pools."""

    def __init__(self, rows: int, cols: int, reduce_size: int = 1):
        super().__init__()
        self.pool_h = nn.Sequential(nn.AvgPool2d((rows, reduce_size)), nn.Flatten())
        self.pool_w = nn.Sequential(nn.AvgPool2d((reduce_size, cols)), nn.Flatten())

    def forward(self, x):
        return torch.cat((self.pool_h(x), self.pool_w(x)), dim=-1)


def image_encoder(units: int = 64, spanning_avg_pool: bool = False) -> nn.Sequential:
    """``observation_image_layers`` (``door_models/layers.py:43-63``,
    ``push_models/layers.py:68-104``); sequential indices as in the reference."""
    trunk = [
        nn.Conv2d(1, 32, kernel_size=5, padding=2), nn.ReLU(),
        ResConv2d(32, kernel_size=3),
        nn.Conv2d(32, 16, kernel_size=3, padding=1), nn.ReLU(),
    ]
    if spanning_avg_pool:
        neck = [nn.Conv2d(16, 2, kernel_size=3, padding=1), DualSpanningAvgPool(32, 32, 2),
                nn.Linear(32 * 2, units)]
    else:
        neck = [nn.Conv2d(16, 8, kernel_size=3, padding=1), nn.Flatten(),
                nn.Linear(8 * 32 * 32, units)]
    return nn.Sequential(*trunk, *neck, nn.ReLU(), ResLinear(units))
