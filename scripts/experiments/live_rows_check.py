"""Does the live-row early exit of the sparse forward kernels bite?  One 64 -> 64 / 16 -> 16 table of 40 000 rows, live count 40 000 / 20 000 / 1 000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, rslo_amd
from rslo_amd import capi
g = torch.Generator().manual_seed(3)
n, K = 40000, 27
nbr = torch.randint(0, n, (n, K), generator=g, dtype=torch.int32)
nbr = torch.where(torch.rand((n, K), generator=g) < 0.4, nbr, torch.full_like(nbr, -1)).cuda()
for cin, cout in ((64, 64), (16, 16), (7, 16)):
    x = torch.randn(n, cin, device="cuda"); W = torch.randn(K, cin, cout, device="cuda") * 0.1
    with capi.tuning(spconv_rbw=1, spconv_ks=4):
        for live in (None, 40000, 20000, 1000):
            nl = None if live is None else torch.tensor([live], dtype=torch.int32, device="cuda")
            f = lambda: capi.spconv_fwd(x, W, None, nbr, n_live=nl)
            for _ in range(5): f()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): f()
            e1.record(); torch.cuda.synchronize()
            print("%d->%d live=%s: %.1f us" % (cin, cout, live, e0.elapsed_time(e1) * 1e3 / 50))
