# HBM traffic counters for the sparse-conv micro-benchmark kernels (one counter family per pass)
OUT=/tmp/pmc_tr; rm -rf $OUT; mkdir -p $OUT gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  REPS=3 ONLY="subm" timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python $GRAFT_REPO_ROOT/scripts/bench_spconv.py > $OUT/$c.log 2>&1
done
python3 - $OUT <<'PY'
import csv, sys, glob, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (sys.argv[1], c), recursive=True)[0]
    agg = collections.defaultdict(list)
    for d in csv.DictReader(open(f)):
        if "spconv" in d["Kernel_Name"] and d["Counter_Name"] == c:
            agg[d["Kernel_Name"].split("(")[0]].append(float(d["Counter_Value"]))
    for k, v in agg.items():
        res[k][c] = sum(v) / len(v)
print("kernel, FETCH_SIZE KB (raw), x2 corrected MB, WRITE_SIZE KB, total MB/launch")
for k, v in sorted(res.items()):
    fs, ws = v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)
    print("%-44s fetch_raw=%10.0f KB  fetch_x2=%8.1f MB  write=%8.1f MB  total=%8.1f MB" % (k, fs, 2 * fs / 1024, ws / 1024, (2 * fs + ws) / 1024))
PY
