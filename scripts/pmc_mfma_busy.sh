# Matrix-core busy fraction of bench.py's MFMA kernels: one PMC pass (kernel-trace only) with SQ_VALU_MFMA_BUSY_CYCLES and
# GRBM_GUI_ACTIVE per dispatch.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (checked: 16 cycles x the MFMA count of
# k_spconv_v6), GRBM_GUI_ACTIVE over the 8 XCDs (GUI_ACTIVE / 8 / duration = 2.4 GHz), so
#   busy % = 100 * MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024).
OUT=/tmp/pmc_mf; rm -rf $OUT; mkdir -p $OUT gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/run.log 2>&1
echo "rocprof rc=$?"; tail -2 $OUT/run.log | cut -c1-200
python3 - $OUT $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r02}_pmc_mfma_busy.json $GRAFT_REPO_ROOT/rslo_amd/librslo_hip.so <<'PY'
import csv, sys, glob, collections, json, re
f = glob.glob("%s/**/*counter_collection.csv" % sys.argv[1], recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in csv.DictReader(open(f)):
    n = d["Kernel_Name"]
    if any(k in n for k in ("k_spconv", "k_wgrad", "k_conv2d_fwd", "k_conv2d_wgrad_s1", "k_conv2d_wgrad<", "k_conv2d_str", "miopenSp3", "igemm")):
        per[re.sub(r"^void ", "", n.split("(")[0])[:60]][d["Counter_Name"]].append(float(d["Counter_Value"]))
import hashlib
out = {"lib_sha256": __import__("runpy").run_path(__import__("os").path.join(__import__("os").path.dirname(sys.argv[3]), "build.py"))["source_hash"](),
       "note": "per-launch averages over a 5-step bench.py run; mfma_busy_pct = 100 * SQ_VALU_MFMA_BUSY_CYCLES / "
               "(GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); shader_clock_cycles = GRBM_GUI_ACTIVE / 8", "kernels": {}}
for k, v in sorted(per.items()):
    mb, ga = v.get("SQ_VALU_MFMA_BUSY_CYCLES", []), v.get("GRBM_GUI_ACTIVE", [])
    if not mb or not ga:
        continue
    m, g = sum(mb) / len(mb), sum(ga) / len(ga)
    out["kernels"][k] = {"launches_sampled": len(mb), "mfma_busy_cycles": round(m), "shader_clock_cycles": round(g / 8),
                         "mfma_busy_pct": round(100.0 * m / (g / 8.0 * 1024.0), 2) if g else None}
    print("%-62s launches %4d  MFMA busy %6.2f %%" % (k, len(mb), out["kernels"][k]["mfma_busy_pct"] or 0))
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
