"""multimodalfilter_amd -- MI355X-native differentiable particle / EKF filtering hot path.

The package mirrors the API surface the reference (brentyi/multimodalfilter) is written
against -- ``base`` / ``filters`` / ``types`` play the role of ``torchfilter``'s modules,
``base_models`` / ``door_models`` / ``push_models`` that of ``crossmodal``'s -- while the
per-particle and per-trajectory-belief arithmetic runs in hand-written HIP
(``csrc/*.hip`` -> ``libmmf_hip.so``, C ABI in ``include/mmf.h``).  Importing the package
does not need a GPU; running a filter does, and there is no CPU fallback.
"""
from . import _abi, base, types, utils  # noqa: F401
from . import filters, base_models  # noqa: F401
from . import door_models, push_models  # noqa: F401
from . import data, train  # noqa: F401
from .utils import CounterNoise, NoiseSource, ReplayNoise, StackedNoise  # noqa: F401

__all__ = ["base", "filters", "types", "utils", "base_models", "door_models", "push_models", "data", "train",
           "NoiseSource", "ReplayNoise", "StackedNoise", "CounterNoise", "model_types"]


def model_types(task: str):
    """Registry of filter classes by reference class name
    (``/root/reference/crossmodal/tasks/_task.py:15-28``)."""
    return {"door": door_models.model_types, "push": push_models.model_types}[task]
