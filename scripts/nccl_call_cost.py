"""Host and device cost of one small torch.distributed all_reduce on a 1-rank RCCL group (the per-call overhead a
SyncBN statistics exchange pays on top of the collective's own latency).  Run: python scripts/nccl_call_cost.py"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.zeros(257, dtype=torch.float64, device="cuda")
y = torch.randn(1 << 20, device="cuda")
for _ in range(20):
    dist.all_reduce(x)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    dist.all_reduce(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"all_reduce(257 f64), 1 rank: host {1e6*(t1-t0)/n:.1f} us per call, drained {1e6*(t2-t0)/n:.1f} us per call")
# the same with a small kernel between the calls (the BN apply), as in the step
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, f in (("kernel only", lambda: y.mul_(1.0)), ("kernel + all_reduce", lambda: (y.mul_(1.0), dist.all_reduce(x)))):
    torch.cuda.synchronize(); e0.record()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: host {1e6*(t1-t0)/n:.1f} us, device timeline {1e3*e0.elapsed_time(e1)/n:.1f} us per iteration")
dist.destroy_process_group()
