"""Minimal apex stand-in for ROCm (used only when NVIDIA apex is not importable).

The reference imports apex for three things on the hot path (Dockerfile:93-97 pins f3a960f8):
`amp.float_function` decorators (fp32 islands), `parallel.SyncBatchNorm` (rslo/layers/SparseConv.py:96)
and `parallel.DistributedDataParallel` (train_hdf5.py:463).  On MI355X these map onto torch-native
pieces running over RCCL; see amp.py / parallel.py.
"""
from . import amp, parallel  # noqa: F401
