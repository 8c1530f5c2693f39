# Copies the outputs of scripts/final_measure.sh from gpurun_out/ (scratch) into profiles/ (tracked) under the names
# profiles/README.md lists.  ROUND=r02 bash scripts/collect_profiles.sh
R=${ROUND:-r02}
G=gpurun_out; P=profiles
# the median of the three plain runs is the bench line of the round
python3 - "$R" <<'PY'
import json, sys
R = sys.argv[1]
runs = [json.loads(open("gpurun_out/bench_%s_%d.json" % (R, i)).read()) for i in (1, 2, 3)]
runs.sort(key=lambda d: d["ms_per_step"])
open("profiles/%s_bench_line.json" % R, "w").write(json.dumps(runs[1]) + "\n")
print("bench line: %.1f frame-pairs/s, %.3f ms per step (runs: %s)" % (runs[1]["value"], runs[1]["ms_per_step"],
      ", ".join("%.3f" % d["ms_per_step"] for d in runs)))
PY
for c in c2 c4 c5; do [ -f $G/bench_${R}_$c.json ] && cp $G/bench_${R}_$c.json $P/${R}_bench_line_$c.json; done
cp $G/bench_${R}_with_cpu.json $P/${R}_bench_line_with_cpu_baseline.json
[ -f $G/bench_${R}_driver_invocation.json ] && cp $G/bench_${R}_driver_invocation.json $P/${R}_bench_line_driver_invocation.json
cp $G/${R}_kernel_stats.csv $P/${R}_kernel_stats.csv
cp $G/prof_${R}_last_step.txt $P/${R}_step_breakdown.txt
cp $G/prof_${R}fix_last_step.txt $P/${R}_step_breakdown_fixed_plan.txt
for f in pmc_mfma_busy pmc_sq_waits pmc_traffic_bench; do cp $G/${R}_$f.json $P/${R}_$f.json; done
for f in timeline_gaps.txt torch_launch_sites.txt nccl_n1.log two_rank_one_gpu.log; do [ -f $G/${R}_$f ] && cp $G/${R}_$f $P/${R}_$f; done
grep -h lib_sha256 $P/${R}_pmc_*.json | sort | uniq -c
