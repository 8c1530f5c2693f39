"""The three torchplus helpers the hot path uses."""
from . import nn, ops, tools  # noqa: F401
from .ops.array_ops import roll  # noqa: F401
from .tools import change_default_args  # noqa: F401
