"""rslo_amd -- MI355X-native implementation of the RSLO two-frame LiDAR-odometry hot path.

Layout
  csrc/ + librslo_hip.so   hand-written HIP kernels (gfx950) behind the C ABI of include/rslo_hip.h
  capi.py                  ctypes binding of that ABI (torch tensors -> device pointers)
  spconv/, thirdparty/, rslo/, torchplus/
                           host-side mirror of the reference's module API for this path, importable
                           under the reference's own top-level names (`import spconv`,
                           `from rslo.models import ...`, `from thirdparty.chamfer_distance...`)
  compat/                  apex / kornia stand-ins, used only when the real packages are absent

Importing this package puts those top-level names on sys.path (the reference asks its users to
put $ROOT and $ROOT/rslo on PYTHONPATH in the same way, README.md:72-73).
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
_COMPAT = os.path.join(_HERE, "compat")
for _name in ("apex", "kornia"):
    if _name not in sys.modules and importlib.util.find_spec(_name) is None and _COMPAT not in sys.path:
        sys.path.append(_COMPAT)

__version__ = "0.1.0"
