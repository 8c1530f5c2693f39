"""rslo_vfe_mean on seeded inputs: writes the outputs to a file (RSLO_TUNING=vfe_lds=0: one thread per voxel from memory, 1: rows
staged through LDS) and compares with another run's file -- the two kernels must agree in every bit.
usage: RSLO_TUNING=vfe_lds=0 python scripts/check_vfe_bits.py a.pt; RSLO_TUNING=vfe_lds=1 python scripts/check_vfe_bits.py b.pt a.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rslo_amd
from rslo_amd import capi
import _tuning; _tuning.apply_from_env()
g = torch.Generator().manual_seed(11)
out = []
for M, T, F in ((125431, 10, 7), (63, 10, 7), (64, 10, 7), (1000, 5, 4), (257, 3, 16), (31496, 10, 7)):
    num = torch.randint(1, T + 1, (M,), generator=g, dtype=torch.int32)
    vox = torch.randn(M, T, F, generator=g) * 10
    vox = vox * (torch.arange(T)[None, :, None] < num[:, None, None])      # empty slots are zeros, like the voxelizer's
    vox, num = vox.cuda().contiguous(), num.cuda()
    m = capi.vfe_mean(vox, num)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        capi.vfe_mean(vox, num)
    e1.record(); torch.cuda.synchronize()
    print("M=%d T=%d F=%d: %.1f us" % (M, T, F, e0.elapsed_time(e1) / 50 * 1e3))
    out.append(m.cpu())
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    same = all(torch.equal(a, b) for a, b in zip(out, torch.load(sys.argv[2])))
    print("identical bits:", same)
    sys.exit(0 if same else 1)
