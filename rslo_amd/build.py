"""Build librslo_hip.so (the C-ABI of include/rslo_hip.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the authoring container; the built
library travels to the GPU box with the snapshot.  No torch involvement: the library has a
plain C ABI and links only the HIP runtime.
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librslo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP pass pairs independent fp32 operations into v_pk_*_f32, which issue at half rate on
# gfx950 and need their operands in adjacent registers (extra v_mov): measured 80 -> 128 us on k_wgrad3<64,64>
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize",
         "-Wno-unused-result"] + os.environ.get("RSLO_HIPCC_EXTRA", "").split()      # (experiments: -DSPC_PREW_ALL=1 ...)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash():
    """First 16 hex digits of the sha256 over what librslo_hip.so is built FROM: every csrc/*.hip and csrc/*.h, include/rslo_hip.h
    and the compiler flags.  The PMC summaries under profiles/ are stamped with it and bench.py quotes them only for the same
    sources (the hash of the binary differs between two builds of identical sources on different boxes)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "rslo_hip.h")]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(HERE, "..", "include", "rslo_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    for s in sources():
        o = os.path.join(HERE, "_obj", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print("[rslo_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out))
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[rslo_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


HOST_SRC = os.path.join(CSRC, "host", "voxelize_host.c")
HOST_SRCS = [HOST_SRC, os.path.join(CSRC, "host", "chamfer_host.c")]
HOST_LIB = os.path.join(HERE, "librslo_host.so")
HOST_FLAGS = ["-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wextra"]


def build_host(force=False, verbose=True):
    """librslo_host.so (include/rslo_host.h): the host-memory face of VoxelGenerator for forked DataLoader workers.
    Plain C through gcc, no HIP."""
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= max(os.path.getmtime(f) for f in HOST_SRCS):
        return HOST_LIB
    cmd = [os.environ.get("CC", "gcc")] + HOST_FLAGS + ["-o", HOST_LIB] + HOST_SRCS + ["-lm"]
    if verbose:
        print("[rslo_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == "__main__":
    print(build(force=True))
    print(build_host(force=True))
