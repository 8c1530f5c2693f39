"""dense() scatter / gather (a8) at the C3 step's size: tiled (dense_tiled = 1) vs one thread per element (0); bits compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
torch.manual_seed(0)
B, C, dims, M = 8, 64, [2, 96, 176], 21072
lin = torch.randperm(B * dims[0] * dims[1] * dims[2], device="cuda")[:M].sort().values
b = lin // (dims[0] * dims[1] * dims[2]); r = lin % (dims[0] * dims[1] * dims[2])
coords = torch.stack([b, r // (dims[1] * dims[2]), (r // dims[2]) % dims[1], r % dims[2]], 1).int().contiguous()
feat = torch.randn(M, C, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
res = {}
for mode in (1, 0):
    with capi.tuning(dense_tiled=mode):
        for frames in (1, 2):
            d = capi.dense_scatter(feat, coords, B, dims, frames=frames)
            g = capi.dense_gather(d, coords, C, B, dims, frames=frames)
            res[(mode, frames)] = (d, g)
            print("tiled=%d frames=%d: scatter (incl. 69 MB fill) %.1f us, gather %.1f us, gather(scatter(x)) == x: %s" % (
                mode, frames, t(lambda: capi.dense_scatter(feat, coords, B, dims, frames=frames)),
                t(lambda: capi.dense_gather(d, coords, C, B, dims, frames=frames)), bool(torch.equal(g, feat))))
print("same bits:", all(torch.equal(res[(1, f)][0], res[(0, f)][0]) and torch.equal(res[(1, f)][1], res[(0, f)][1]) for f in (1, 2)))
