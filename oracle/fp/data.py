"""Placeholder for ``fannypack.data`` (Drive download + HDF5: out of scope)."""


def cached_drive_file(*_a, **_k):
    raise RuntimeError("datasets are not available offline (SURVEY.md section 2, row 11)")


class TrajectoriesFile:
    def __init__(self, *_a, **_k):
        raise RuntimeError("HDF5 trajectories are not available offline")
