"""What a FIFTH stream in the process costs the C3 step (round 6: a 24-byte pinned upload per step on a stream of its own ran the
step at half speed).  MODE = none | idle (stream created, unused) | kernel (one tiny kernel per step on it) | kernel2 (on two further streams) | copy (one pinned
H2D copy per step on it) | copy_wait (copy + the training stream waits for its event) | copy_wait_plan (the same on the
structure stream instead of a new one).  PRIO = priority of the new stream (-1 high, 0, 1 low).  Wraps ExamplePrefetcher.get and runs bench.py in this process."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, rslo_amd  # noqa: E402
from rslo_amd import workload  # noqa: E402
MODE = os.environ.get("MODE", "none")
state = {}
orig_get = workload.ExamplePrefetcher.get


def get(self):
    ex = orig_get(self)
    if MODE == "none":
        return ex
    if "s" not in state:
        state["s"] = self.stream if MODE == "copy_wait_plan" else torch.cuda.Stream(self.device, priority=int(os.environ.get("PRIO", "0")))
        state["buf"] = torch.zeros(8, device=self.device)
    s = state["s"]
    cur = torch.cuda.current_stream(self.device)
    if MODE in ("kernel", "kernel2"):
        with torch.cuda.stream(s):
            state["buf"].add_(1.0)
        if MODE == "kernel2":      # and a second extra stream
            if "s2" not in state:
                state["s2"] = torch.cuda.Stream(self.device)
            with torch.cuda.stream(state["s2"]):
                state["buf"].add_(1.0)
    elif MODE in ("copy", "copy_wait", "copy_wait_plan"):
        with torch.cuda.stream(s):
            dev = torch.arange(6, dtype=torch.int32).pin_memory().to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(s)
        if MODE != "copy":
            dev.record_stream(cur)
            cur.wait_event(ev)
        state["keep"] = dev
    return ex


workload.ExamplePrefetcher.get = get
os.environ["RSLO_BENCH_MULTIRANK_CHILD"] = "0"
os.environ["RSLO_BENCH_OTHER_CONFIGS"] = "0"
sys.argv = [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "40", "--warmup", "10"]
runpy.run_path(sys.argv[0], run_name="__main__")
