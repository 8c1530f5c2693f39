"""Post-process a rocprofv3 kernel-trace CSV: for the last steady-state step (window between two k_conv2d_wsplit_many
launches) print per queue: busy time, idle gaps, and how the busy time splits by launch size (workgroups per launch)."""
import csv, sys, collections
path = sys.argv[1]
rows = []
with open(path) as f:
    for d in csv.DictReader(f):
        wg = max(1, int(d.get("Workgroup_Size_X", d.get("Workgroup_Size", "1")) or 1))
        grid = int(d.get("Grid_Size_X", d.get("Grid_Size", "0")) or 0) * max(1, int(d.get("Grid_Size_Y", "1") or 1))
        wgy = max(1, int(d.get("Workgroup_Size_Y", "1") or 1))
        rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"], d.get("Queue_Id", "0"),
                     grid // (wg * wgy) if grid else 0))
rows.sort()
marks = [s for s, e, n, q, g in rows if n.startswith("k_conv2d_wsplit_many")]
t0, t1 = marks[-3], marks[-2]
sel = [r for r in rows if t0 <= r[0] < t1]
print("window %.3f ms, %d dispatches" % ((t1 - t0) / 1e6, len(sel)))
byq = collections.defaultdict(list)
for r in sel:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, *_ in rs) / 1e6
    gaps = [(rs[i + 1][0] - rs[i][1]) for i in range(len(rs) - 1)]
    gap_pos = sum(g for g in gaps if g > 0) / 1e6
    print("queue %s: %d launches, busy %.3f ms, idle between launches %.3f ms (gaps > 5 us: %d, sum %.3f ms)" % (
        q, len(rs), busy, gap_pos, sum(1 for g in gaps if g > 5000), sum(g for g in gaps if g > 5000) / 1e6))
    buckets = collections.OrderedDict([("<64 wg", 0.0), ("64-255", 0.0), ("256-1023", 0.0), ("1024-4095", 0.0), (">=4096", 0.0)])
    cnt = collections.Counter()
    for s, e, n, _, g in rs:
        k = "<64 wg" if g < 64 else "64-255" if g < 256 else "256-1023" if g < 1024 else "1024-4095" if g < 4096 else ">=4096"
        buckets[k] += (e - s) / 1e6
        cnt[k] += 1
    print("   busy by launch size:", {k: (cnt[k], round(v, 3)) for k, v in buckets.items()})
    allk = collections.defaultdict(lambda: [0, 0.0])
    for s_, e_, n_, _, g_ in rs:
        allk[n_[:60]][0] += 1
        allk[n_[:60]][1] += (e_ - s_) / 1e6
    if len(rs) < 400:      # a side queue: every kernel
        print("   kernels of this queue:")
        for n_, (c_, t_) in sorted(allk.items(), key=lambda kv: -kv[1][1])[:30]:
            print("      %4d %7.3f ms %7.1f us  %s" % (c_, t_, 1e3 * t_ / c_, n_))
    else:                  # the training queue: by launch count (each launch also costs a dispatch gap)
        print("   kernels of this queue by launch count:")
        for n_, (c_, t_) in sorted(allk.items(), key=lambda kv: -kv[1][0])[:70]:
            print("      %4d %7.3f ms %7.1f us  %s" % (c_, t_, 1e3 * t_ / c_, n_))
    small = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, _, g in rs:
        if g < 256:
            small[n[:60]][0] += 1
            small[n[:60]][1] += (e - s) / 1e6
    for n, (c, t) in sorted(small.items(), key=lambda kv: -kv[1][1])[:14]:
        print("      %4d %7.3f ms  %s" % (c, t, n))

# launches that hold the stream with very few workgroups (latency-bound chains): candidates for restructuring
print("== launches with < 64 workgroups lasting > 12 us (all queues), per step")
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for s, e, n, q, g in sel:
    if g < 64 and (e - s) > 12000:
        a = agg[n[:70]]
        a[0] += 1; a[1] += (e - s) / 1e3; a[2] = max(a[2], g)
for n, (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("   %3d x %7.1f us avg (<= %3d wg)  %s" % (c, t / c, g, n))

# SEG_FROM / SEG_TO (kernel-name prefixes): every launch of the training queue between the first SEG_FROM and the first
# SEG_TO after it -- start offset, duration, gap to the previous launch (the serial stretch between the head's forward and the
# backward pass is SEG_FROM=k_pair_rows_fwd SEG_TO=k_head_masks_bwd)
import os
if os.environ.get("SEG_FROM"):
    fr, to = os.environ["SEG_FROM"], os.environ.get("SEG_TO", "")
    mainq = max(byq.items(), key=lambda kv: len(kv[1]))[0]
    rs = byq[mainq]
    i0 = next((i for i, r in enumerate(rs) if r[2].startswith(fr)), None)
    if i0 is not None:
        i1 = next((i for i in range(i0 + 1, len(rs)) if to and rs[i][2].startswith(to)), len(rs) - 1)
        base, busy, prev_end = rs[i0][0], 0, rs[i0][0]
        print("== training queue from %s to %s: %d launches" % (fr, to, i1 - i0 + 1))
        for s, e, n, _, g in rs[i0:i1 + 1]:
            print("   +%8.1f us  %7.1f us  gap %6.1f  wg %6d  %s" % ((s - base) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g, n[:70]))
            busy += e - s
            prev_end = e
        print("   span %.1f us, busy %.1f us" % ((rs[i1][1] - base) / 1e3, busy / 1e3))
