"""Host-side mirror of the reference's `rslo` package, restricted to the hot path
(SURVEY.md section 8): same module paths, class names, constructor kwargs and state-dict keys.

Co-existence with a reference checkout.  This package shadows the reference's `rslo`, but only
re-implements the modules ON the hot path.  Everything else the reference's drivers import -- `rslo.protos.*`,
`rslo.utils.log_tool`, `rslo.utils.progress_bar`, `rslo.utils.util`, `rslo.builder.input_reader_builder`,
`rslo.builder.dataset_builder`, the dataset readers (train_hdf5.py:19-36) -- is the maintainer's own copy: when a
reference checkout is on PYTHONPATH (README.md:72-73 asks for $ROOT there) or named by RSLO_REFERENCE_ROOT,
  * every mirror (sub)package appends the checkout's same-named directory to its `__path__`, so modules the mirror
    does not have resolve to the checkout's files, and
  * mirror modules that cover only part of their namesake forward unknown attributes to the checkout's file
    (`reference_fallback`, PEP 562), so `from rslo.data.dataset import get_dataset_class` keeps working.
Mirror modules always win; nothing is copied.
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_root_cache = []


def reference_root():
    """Directory of a reference checkout (the one holding rslo/protos/pipeline_pb2.py), or None."""
    if _root_cache:
        return _root_cache[0]
    found = None
    for cand in [os.environ.get("RSLO_REFERENCE_ROOT")] + list(sys.path):
        if not cand:
            continue
        d = os.path.join(cand, "rslo")
        if os.path.isfile(os.path.join(d, "protos", "pipeline_pb2.py")) and os.path.abspath(d) != _HERE:
            found = os.path.abspath(cand)
            break
    _root_cache.append(found)
    return found


def extend_path(path, name):
    """Append the checkout's directory of package `name` ("rslo.utils") to that package's __path__."""
    root = reference_root()
    if root is None:
        return
    d = os.path.join(root, *name.split("."))
    if os.path.isdir(d) and d not in path:
        path.append(d)


def reference_fallback(modname):
    """-> module-level __getattr__ forwarding names this mirror module lacks to the checkout's same-named file
    (loaded once under `<modname>.__reference__`)."""
    def __getattr__(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        key = modname + ".__reference__"
        ref = sys.modules.get(key)
        if ref is None:
            root = reference_root()
            f = None if root is None else os.path.join(root, *modname.split(".")) + ".py"
            if f is None or not os.path.isfile(f):
                raise AttributeError("module %r has no attribute %r (outside the RSLO hot path, and no reference "
                                     "checkout on PYTHONPATH / RSLO_REFERENCE_ROOT to take it from)" % (modname, attr))
            spec = importlib.util.spec_from_file_location(key, f)
            ref = importlib.util.module_from_spec(spec)
            sys.modules[key] = ref
            try:
                spec.loader.exec_module(ref)
            except BaseException:
                del sys.modules[key]
                raise
        return getattr(ref, attr)
    return __getattr__


extend_path(__path__, __name__)
