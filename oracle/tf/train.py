"""Train-loop names referenced by ``crossmodal/train_helpers.py:45,71,93,116,155`` (out of scope)."""


def _unavailable(*_a, **_k):
    raise RuntimeError("torchfilter.train loops are out of scope (SURVEY.md #17)")


train_dynamics_single_step = train_dynamics_recurrent = _unavailable
train_particle_filter_measurement = train_virtual_sensor = train_filter = _unavailable
