# Where the waves of bench.py's MFMA kernels spend their cycles: SQ wait / active breakdown (one PMC pass, kernel-trace only).
OUT=/tmp/pmc_sq; rm -rf $OUT; mkdir -p $OUT gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $OUT -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/run.log 2>&1
echo "rocprof rc=$?"; tail -2 $OUT/run.log | cut -c1-160
python3 - $OUT $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r02}_pmc_sq_waits.json <<'PY'
import csv, sys, glob, collections, json, re
f = glob.glob("%s/**/*counter_collection.csv" % sys.argv[1], recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in csv.DictReader(open(f)):
    n = d["Kernel_Name"]
    if any(k in n for k in ("k_spconv_v6", "k_wgrad3", "k_conv2d_fwd", "k_conv2d_wgrad_s1", "k_conv2d_str", "k_cg_search")):
        per[re.sub(r"^void ", "", n.split("(")[0])[:60]][d["Counter_Name"]].append(float(d["Counter_Value"]))
out = {"note": "per-launch averages; fractions of SQ_WAVE_CYCLES (quad-cycles summed over waves): wait_any = parked on "
               "s_waitcnt / barrier, wait_inst = issue stalls, active = issuing", "kernels": {}}
for k, v in sorted(per.items()):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    row = {c: round(a.get(c, 0) / wc, 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS",
                                                    "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS")}
    row["waves"] = round(a.get("SQ_WAVES", 0))
    out["kernels"][k] = row
    print("%-30s" % k, row)
import hashlib, os
out["lib_sha256"] = __import__("runpy").run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "rslo_amd/build.py"))["source_hash"]()
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
