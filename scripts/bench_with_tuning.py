"""A/B runs of bench.py under explicit launch-code switches: RSLO_TUNING="name=value,..." python scripts/bench_with_tuning.py <bench args>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import rslo_amd  # noqa: F401
import _tuning
print("tuning:", _tuning.apply_from_env(), file=sys.stderr)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
