from . import resblocks  # noqa: F401
