"""Compile the reference's own chamfer CPU path into oracle/_ref/ (TEST INFRASTRUCTURE).

Source: /root/reference/thirdparty/chamfer_distance/chamfer_distance.cpp, compiled with
g++ directly from where it lies (no copy into this repo, no reference build system, no
stand-in sources).  The four CUDA launchers the file declares (chamfer_distance.cpp:4-46)
stay *undefined*: a shared object may carry undefined function symbols, and the module is
imported with RTLD_LAZY so they are never resolved -- only the CPU entry points
`forward` / `backward` (chamfer_distance.cpp:147-234) are ever called.

Runs only where /root/reference exists (the authoring container).  The GPU box uses the
prebuilt oracle/_ref/*.so that travels with the snapshot (oracle/_ref/ is git-ignored,
not gpurun-ignored).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/thirdparty/chamfer_distance/chamfer_distance.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
NAME = "cd_ref"


def build(verbose=True):
    if not os.path.exists(SRC):
        if verbose:
            print("[oracle/_ref] reference tree absent; keeping prebuilt files")
        return None
    import torch
    from torch.utils.cpp_extension import include_paths

    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, NAME + sysconfig.get_config_var("EXT_SUFFIX"))
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(SRC):
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    incs = []
    for p in include_paths():
        incs += ["-isystem", p]
    incs += ["-isystem", sysconfig.get_paths()["include"]]
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w",
           "-DTORCH_EXTENSION_NAME=" + NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           *incs, SRC, "-o", out,
           "-L" + tlib, "-Wl,-rpath," + tlib, "-Wl,-z,lazy",
           "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    if verbose:
        print("[oracle/_ref]", " ".join(cmd))
    subprocess.check_call(cmd)
    return out


def load():
    """Import the prebuilt reference module (lazy binding; see module docstring)."""
    import importlib.util
    import glob
    import torch  # noqa: F401  (libtorch must be loaded first)
    cands = glob.glob(os.path.join(OUT_DIR, NAME + "*.so"))
    if not cands:
        return None
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location(NAME, cands[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    return mod


if __name__ == "__main__":
    print(build())
