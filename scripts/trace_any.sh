# usage: trace_any.sh <script.py> [pattern]  -> per-kernel stats
OUT=/tmp/ta; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o m -- python $GRAFT_REPO_ROOT/$1 > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" "${2:-}" <<'PY'
import csv, sys
for d in list(csv.DictReader(open(sys.argv[1])))[:60]:
    if sys.argv[2] and sys.argv[2] not in d["Name"]: continue
    print("%6s calls %10.1f us avg (min %8.1f max %8.1f) %s" % (d["Calls"], float(d["AverageNs"]) / 1e3, float(d["MinNs"]) / 1e3, float(d["MaxNs"]) / 1e3, d["Name"][:90]))
PY
