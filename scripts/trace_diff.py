"""Per-kernel GPU time of the last steps of two rocprofv3 kernel traces, side by side (which launches get slower when
another queue is busy).  usage: trace_diff.py A_kernel_trace.csv B_kernel_trace.csv [steps]"""
import csv, sys, collections

def load(path, steps):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # steps are delimited by the optimiser launch
    ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_opt_adam")]
    ends = ends[-(steps + 1):]
    sel = rows[ends[0] + 1: ends[-1] + 1]
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        k = r["Kernel_Name"][:70]
        tot[k][0] += 1
        tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = len(ends) - 1
    span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3 / n
    return {k: (c / n, t / n) for k, (c, t) in tot.items()}, span

steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
a, sa = load(sys.argv[1], steps)
b, sb = load(sys.argv[2], steps)
print("step span us: A %.1f  B %.1f" % (sa, sb))
print("sum busy us : A %.1f  B %.1f" % (sum(t for _, t in a.values()), sum(t for _, t in b.values())))
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
for k in keys[:45]:
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    print("%7.1f %7.1f  %+7.1f  (%4.1f / %4.1f)  %s" % (ta, tb, tb - ta, ca, cb, k))
