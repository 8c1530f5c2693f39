# kernel-trace of an arbitrary script ($SCRIPT, default bench_chamfer2.py): per-kernel avg durations
OUT=/tmp/ta; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 ${LIMIT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o m -- python $GRAFT_REPO_ROOT/scripts/${SCRIPT:-bench_chamfer2.py} > $OUT/log.txt 2>&1
tail -${TAIL:-8} $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for d in rows[:24]:
    print("%6s calls %10.1f us avg %9.1f us total  %s" % (d["Calls"], float(d["AverageNs"]) / 1e3, float(d["TotalDurationNs"]) / 1e3, d["Name"][:100]))
PY
