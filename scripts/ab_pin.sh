# A/B: bench.py with and without pinning the process to an idle core block (alternating runs on one box)
for i in 1 2 3 4; do for pin in 0 1; do
  RSLO_BENCH_PIN=$pin timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab.json
  python - $pin <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); c = d["config"]
print("pin", sys.argv[1], d["ms_per_step"], "host", c["host_issue_ms_per_step"], "timeline", c["ms_per_step_gpu_timeline"]["median"], c.get("pinned_cpus"))
PY
done; done
