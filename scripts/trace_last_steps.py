"""Post-process a rocprofv3 kernel trace CSV: per-kernel stats restricted to n steady-state steps of bench.py
(a step starts with the first k_vox_insert of its 2*batch voxelizer launches).  The window is the n steps BEFORE
the last one, [start of step -(n+1), start of the last step): the last step's tail would include bench.py's
post-timing summary work (pair counting for the roofline figures).  Writes a small text summary."""
import csv
import os
import sys
from collections import defaultdict

path, n_steps, vox_per_step = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out = sys.argv[4] if len(sys.argv) > 4 else None
rows = []
with open(path) as f:
    r = csv.DictReader(f)
    for d in r:
        rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"]))
rows.sort()
vox = [s for s, e, n in rows if n.startswith(os.environ.get("STEP_MARK", "k_vox_insert"))]   # first kernel of a step
assert len(vox) >= (n_steps + 1) * vox_per_step, (len(vox), n_steps, vox_per_step)
t0 = vox[-(n_steps + 1) * vox_per_step]
t1 = vox[-vox_per_step]
sel = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
wall = (t1 - t0) / 1e6
agg = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    a = agg[n]
    a[0] += 1
    a[1] += e - s


def cat(n):
    if n.startswith("k_") or n.startswith("void k_"):
        return "rslo_hip (hand-written)"
    if "naive_conv" in n:
        return "MIOpen naive conv"
    if n.startswith("miopen") or "igemm" in n or "Im2d2Col" in n or "Col2Im" in n or "ck::" in n or "_ZN2ck" in n \
            or "batched_transpose" in n or "SubTensorOp" in n or "MIOpen" in n:
        return "MIOpen / CK conv"
    if n.startswith("Cijk_"):
        return "rocBLAS / hipBLASLt GEMM"
    if "at::native" in n or "at_cuda" in n:
        return "torch elementwise / reduce"
    return "other"


cats = defaultdict(float)
for n, (c, t) in agg.items():
    cats[cat(n)] += t / 1e6
busy = sum(cats.values())
lines = []
lines.append("# %d steady-state step(s): wall %.2f ms/step, GPU busy %.2f ms/step, %d dispatches/step" %
             (n_steps, wall / n_steps, busy / n_steps, len(sel) // n_steps))
lines.append("## by category (ms/step)")
for k, v in sorted(cats.items(), key=lambda kv: -kv[1]):
    lines.append("%-32s %9.3f" % (k, v / n_steps))
lines.append("## top kernels: calls/step, total ms/step, avg us")
import os
TOP = int(os.environ.get("TOP", "60"))       # TOP=0: every kernel
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:TOP or None]:
    lines.append("%6.1f %9.3f %10.1f  %s" % (c / n_steps, t / 1e6 / n_steps, t / c / 1e3, n[:150]))
detail = os.environ.get("DETAIL")        # e.g. DETAIL=miopenSp3AsmConv: every call's duration (us) in launch order
if detail:
    for key in detail.split(","):
        d = [(e - s_) / 1e3 for s_, e, n in sel if key in n]
        lines.append("## calls of *%s* in launch order (us): %s" % (key, " ".join("%.0f" % v for v in d)))
txt = "\n".join(lines)
print(txt)
if out:
    open(out, "w").write(txt + "\n")
