"""rslo_amd -- MI355X-native implementation of the RSLO two-frame LiDAR-odometry hot path.

Layout
  csrc/ + librslo_hip.so   hand-written HIP kernels (gfx950) behind the C ABI of include/rslo_hip.h
  capi.py                  ctypes binding of that ABI (torch tensors -> device pointers)
  spconv/, thirdparty/, rslo/, torchplus/
                           host-side mirror of the reference's module API for this path, importable
                           under the reference's own top-level names (`import spconv`,
                           `from rslo.models import ...`, `from thirdparty.chamfer_distance...`)
  compat/                  apex / kornia stand-ins, used only when the real packages are absent

Importing this package makes exactly those four top-level names resolve to the mirror (the reference asks its
users to put $ROOT and $ROOT/rslo on PYTHONPATH for the same purpose, README.md:72-73).  The package directory itself
is NOT put on sys.path: `capi`, `build`, `workload`, ... stay importable only as `rslo_amd.<name>`, so they cannot
shadow unrelated installed packages or be loaded twice under two names.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_MIRRORED = ("spconv", "rslo", "torchplus", "thirdparty")
_COMPAT = os.path.join(_HERE, "compat")


class _MirrorFinder(importlib.abc.MetaPathFinder):
    """Top-level `spconv` / `rslo` / `torchplus` / `thirdparty` -> the directories next to this file (ahead of anything
    installed); `apex` / `kornia` -> compat/ only when the real package is absent.  Submodules follow the packages'
    own __path__ (which rslo/__init__.py may extend with a reference checkout)."""

    def find_spec(self, name, path=None, target=None):
        if path is not None:
            return None
        if name in _MIRRORED:
            return importlib.machinery.PathFinder.find_spec(name, [_HERE])
        if name in ("apex", "kornia"):
            rest = [f for f in sys.meta_path if f is not self]
            for f in rest:
                try:
                    spec = f.find_spec(name, None, None)
                except Exception:
                    spec = None
                if spec is not None:
                    return spec
            return importlib.machinery.PathFinder.find_spec(name, [_COMPAT])
        return None


if not any(isinstance(f, _MirrorFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _MirrorFinder())

__version__ = "0.1.0"
