"""Minimal restatement of the ``fannypack`` names the reference touches.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  fannypack is an un-pinned PyPI
dependency of the reference (``/root/reference/setup.py:13``) and is absent here; only
the pieces on the filter path are restated: ``nn.resblocks`` (use sites
``crossmodal/door_models/layers.py:24,40,55,62,79,95``) and
``utils.SliceWrapper / to_torch / to_numpy`` (``crossmodal/eval_helpers.py:55,88-106,140``).
``utils.Buddy`` and ``data`` are placeholders so ``import crossmodal`` resolves.
"""
from . import data, nn, utils  # noqa: F401
