from . import array_ops  # noqa: F401
