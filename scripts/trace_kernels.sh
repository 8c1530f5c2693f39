# usage: trace_kernels.sh <script.py> <kernel substring>  -- median rocprofv3 durations per kernel name + grid
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tk
rocprofv3 --kernel-trace --output-format csv -d /tmp/tk -o p -- python $GRAFT_REPO_ROOT/$1 > /tmp/tk.log 2>&1
python3 - "$(find /tmp/tk -name '*kernel_trace.csv' | head -1)" "$2" <<'PY'
import csv, sys, statistics, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:40], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items(): print("%-42s grid %-8s n=%3d median %.1f us" % (k[0], k[1], len(v), statistics.median(v)))
PY
