# kernel-trace of the sparse-conv micro-benchmark: per-kernel avg durations
OUT=/tmp/tm; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=3 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o m -- python $GRAFT_REPO_ROOT/scripts/bench_spconv.py > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for d in rows[:40]:
    print("%6s calls %10.1f us avg %9.1f us total  %s" % (d["Calls"], float(d["AverageNs"]) / 1e3, float(d["TotalDurationNs"]) / 1e3, d["Name"][:110]))
PY
