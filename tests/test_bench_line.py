"""bench.py end to end on the GPU: the JSON line the driver parses must come out, with the roofline and host-baseline
objects filled (a changed C-ABI signature that the bench's launch probe does not follow shows up here, not at round end)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline(hip):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["steps"] == 3 and line["n_gpus"] == 1 and line["value"] > 0
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and r["achieved"] > 0 and r["peak"] > 0
    c = line["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference")
    assert "workload" in line["config"]
