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


def _last_json(stdout):
    return json.loads([ln for ln in stdout.strip().splitlines() if ln.startswith("{")][-1])


def test_bench_gpus_flag_starts_that_many_ranks_by_itself():
    """`python bench.py --gpus 2` with NO launcher around it (the form the driver uses) must start two ranks itself
    (the reference: train_hdf5.py:255 mp.spawn) and report n_gpus 2.  --dry-run: launcher + gloo rendezvous + the
    barrier / max-over-ranks timing protocol on the CPU, no kernels."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--dry-run"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                       # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["dry_run"] is True
    assert line["ranks"]["world"] == 2 and line["ranks"]["ms_per_step_min"] <= line["ranks"]["ms_per_step_max"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_bench_gpus_2_runs_the_full_two_rank_step(hip):
    """The complete two-rank step (SyncBN statistics exchange, overlapped gradient all-reduce, per-rank prefetch) started
    by `bench.py --gpus 2` itself.  The test boxes have ONE GPU: RSLO_BENCH_ONE_GPU=1 puts both ranks on it over gloo --
    functional, not a performance mode."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["RSLO_BENCH_ONE_GPU"] = "1"
    env["RSLO_PEER_TIMEOUT_MS"] = "20000"          # a peer that does not show up fails the test in seconds, not minutes
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--batch", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["ms_per_step_min"] <= line["rccl"]["ms_per_step_max"]
    # SyncBN statistics travelled through the same-stream peer kernel, and after the steps both ranks hold the same bits
    assert "peer kernel" in line["rccl"]["syncbn_exchange"], line["rccl"]
    assert line["rccl"]["replicas_identical"] is True, line["rccl"]
    # diagnostics of an N > 1 run: the exposed part of the gradient exchange, per-rank transport and per-exchange waits
    r = line["rccl"]
    assert r["grad_exchange_exposed_ms_per_step"] is not None and r["grad_exchange_exposed_ms_per_step"] >= 0
    assert len(r["syncbn_transport_per_rank"]) == 2 and len(set(r["syncbn_transport_per_rank"])) == 1
    assert all(w["samples"] > 0 and w["p50_us"] <= w["p99_us"] <= w["max_us"] for w in r["syncbn_wait_us"]), r["syncbn_wait_us"]
    assert "topology_lines" in r          # RCCL's own INIT / GRAPH lines at N > 1 over RCCL (None in the gloo one-GPU mode)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["c2", "c5"])
def test_bench_encoder_configs_print_the_same_schema(hip, cfg):
    """BASELINE configs[1] / configs[4] through the entry point the driver calls: same JSON schema as the C3 line, roofline from
    the launch probe, CPU baseline over the oracle backend."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "5", "--warmup", "3"],
                         capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["unit"] == "frame-pairs/s"
    assert line["config"]["workload"].startswith(cfg.upper()) and line["config"]["voxelize_and_plan_on_side_stream"] is True
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and "k_spconv" in r["kernel"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
