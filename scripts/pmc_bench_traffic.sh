# HBM traffic (FETCH_SIZE, WRITE_SIZE; one counter per pass, kernel-trace only) of bench.py's sparse-conv kernels
OUT=/tmp/pmc_bt; rm -rf $OUT; mkdir -p $OUT gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/$c.log 2>&1
done
python3 - $OUT $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r02}_pmc_traffic_bench.json $GRAFT_REPO_ROOT/rslo_amd/librslo_hip.so <<'PY'
import csv, sys, glob, collections, json, re
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (sys.argv[1], c), recursive=True)[0]
    agg = collections.defaultdict(list)
    for d in csv.DictReader(open(f)):
        n = d["Kernel_Name"]
        if ("k_spconv" in n or "k_wgrad2<" in n or "k_wgrad3<" in n or "k_conv2d" in n or "k_weight_split" in n or "k_chamfer_part" in n or "k_cg_search" in n) and d["Counter_Name"] == c:
            agg[re.sub(r"^void ", "", n.split("(")[0])].append(float(d["Counter_Value"]))
    for k, v in agg.items():
        res[k][c] = sum(v) / len(v); res[k]["launches"] = len(v)
import hashlib
out = {"lib_sha256": __import__("runpy").run_path(__import__("os").path.join(__import__("os").path.dirname(sys.argv[3]), "build.py"))["source_hash"](),
       "note": "per-launch averages over a 5-step bench.py run; FETCH_SIZE/WRITE_SIZE are reported in KB; "
               "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B)",
       "kernels": {}}
for k, v in sorted(res.items()):
    fs, ws = v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)
    out["kernels"][k] = {"fetch_size_kb_raw": round(fs, 1), "write_size_kb": round(ws, 1), "launches_sampled": v["launches"],
                         "hbm_bytes_per_launch": int((2 * fs + ws) * 1024)}
    print("%-46s fetch_raw %9.0f KB write %9.0f KB -> %7.1f MB/launch" % (k, fs, ws, (2 * fs + ws) / 1024))
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
