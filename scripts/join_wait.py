"""From a rocprofv3 kernel trace of bench.py: per step, when does each queue finish its work relative to the optimizer's
first launch (k_opt_sqnorm) -- i.e. does the training stream wait for the weight-gradient stream at the end of backward?"""
import csv, sys, collections
rows = []
for d in csv.DictReader(open(sys.argv[1])):
    rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"], d.get("Queue_Id", "0")))
rows.sort()
opts = [i for i, r in enumerate(rows) if r[2].startswith("k_opt_sqnorm")]
mainq = rows[opts[-1]][3]
for oi in opts[-6:-1]:
    t_opt = rows[oi][0]
    # last kernel on each queue that ENDS before the optimizer starts, within 15 ms before it
    last = {}
    for s, e, n, q in rows:
        if t_opt - 15_000_000 < s < t_opt and e <= t_opt + 1000:
            if q not in last or e > last[q][0]:
                last[q] = (e, n[:44])
    print("optimizer starts; last end before it, per queue (us before the optimizer's start):")
    for q, (e, n) in sorted(last.items(), key=lambda kv: -kv[1][0]):
        print("   queue %s%s: %8.1f us  %s" % (q, " (training)" if q == mainq else "", (t_opt - e) / 1e3, n))
