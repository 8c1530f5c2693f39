"""N streams, each issuing a chain of K medium kernels per round with a cross-stream event dependency on stream 0 (as the step's
side streams have): wall time per round against N.  Does the runtime serialise beyond four active streams?"""
import os, sys, time, torch
N = int(sys.argv[1]); K = 40; ROUNDS = 30
dev = torch.device("cuda")
streams = [torch.cuda.Stream(dev, priority=-1 if i == 0 else 0) for i in range(N)]
bufs = [torch.randn(1 << 20, device=dev) for _ in range(N)]
def round_():
    ev = torch.cuda.Event(); ev.record(streams[0])
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            if i: s.wait_event(ev)
            for _ in range(K): bufs[i].mul_(1.0001)
    for s in streams[1:]:
        e = torch.cuda.Event(); e.record(s); streams[0].wait_event(e)
for _ in range(3): round_()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(ROUNDS): round_()
torch.cuda.synchronize()
print("streams %d: %.3f ms per round (%d kernels per stream)" % (N, 1e3 * (time.perf_counter() - t0) / ROUNDS, K))
