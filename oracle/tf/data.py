"""Dataset names referenced by ``crossmodal/train_helpers.py:39,63,83`` (out of scope)."""


class _Unavailable:
    def __init__(self, *_a, **_k):
        raise RuntimeError("torchfilter.data datasets are out of scope (SURVEY.md #17)")


SingleStepDataset = SubsequenceDataset = ParticleFilterMeasurementDataset = _Unavailable
