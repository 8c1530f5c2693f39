"""Writes the outputs of rslo_segbn_fwd / rslo_segbn_bwd on seeded ragged inputs to a file (run once per library build
with RSLO_HIP_LIB=..., then compare the files): used to check that a restructured finish kernel keeps every bit.
usage: python scripts/check_segbn_bits.py out.pt [other.pt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
out = {}
g = torch.Generator().manual_seed(3)
for C, lens in ((64, [30011, 29875, 31002, 0, 1, 28000, 30500, 29999]), (32, [7001, 6800, 7123]), (16, [250, 256, 257, 1, 700]),
                (7, [1000, 900])):
    S, n = len(lens), sum(lens)
    off = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32).cuda()
    x = torch.randn(n, C, generator=g).cuda() * 2 + 0.5
    gy = torch.randn(n, C, generator=g).cuda()
    gamma, beta = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    y, mean, invstd = capi.segbn_fwd(x, off, S, max(lens), gamma, beta, rm, rv, 0.01, 1e-3, 0.1)
    gx, dg, db = capi.segbn_bwd(x, y, gy, off, S, max(lens), gamma, mean, invstd, 0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        capi.segbn_fwd(x, off, S, max(lens), gamma, beta, rm.clone(), rv.clone(), 0.01, 1e-3, 0.1)
    torch.cuda.synchronize()
    print("C=%d S=%d: %.1f us per forward (3 launches + 2 clones)" % (C, S, (time.perf_counter() - t0) / 50 * 1e6))
    live = torch.tensor([l > 0 for l in lens]).cuda()      # statistics rows of empty segments are never written
    out[C] = [t.cpu() for t in (y, mean[live], invstd[live], rm, rv, gx, dg, db)]
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    other = torch.load(sys.argv[2])
    same = all(torch.equal(a, b) for C in out for a, b in zip(out[C], other[C]))
    print("identical bits:", same)
    sys.exit(0 if same else 1)
