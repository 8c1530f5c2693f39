"""Structure kernels of one batch (the side-stream work of a step): site hash, SubM / strided rulebooks, pair lists, row
orders -- device time per call at the encoder's level geometry (8 clouds of the synthetic batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rslo_amd
from rslo_amd import capi, synthetic as S, workload


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


clouds = [c for pair in workload.kitti_pairs(4) for c in pair]
cs = []
for b, p in enumerate(clouds):
    v, c, n, nv = capi.voxelize(torch.from_numpy(np.ascontiguousarray(p, np.float32)).cuda(), S.PC_RANGE, S.VOXEL_SIZE,
                                (1408, 768, 40), S.MAX_POINTS_PER_VOXEL, 80000)
    k = int(nv.item())
    cs.append(torch.cat([torch.full((k, 1), b, dtype=torch.int32, device="cuda"), c[:k]], 1))
coords = torch.cat(cs).contiguous()
dims = [41, 768, 1408]
idx = capi.SiteIndex(coords, len(clouds), dims)
print("level 0: %d sites" % coords.shape[0])
for lvl in range(3):
    n = idx.coords.shape[0]
    print("L%d %7d rows | hash %6.1f us" % (lvl, n, timeit(lambda: capi.SiteIndex(idx.coords, idx.batch, idx.dims))), end="")
    nbr = capi.rulebook_subm(idx, [3, 3, 3])
    print(" | subm %6.1f" % timeit(lambda: capi.rulebook_subm(idx, [3, 3, 3])), end="")
    print(" | pairs(subm) %6.1f" % timeit(lambda: capi.rulebook_pairs(nbr)), end="")
    oidx, nb, nbT = capi.rulebook_conv(idx, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    print(" | conv (count+emit+hash+nbr+nbrT, 1 host read) %6.1f" % timeit(lambda: capi.rulebook_conv(idx, [3, 3, 3], [2, 2, 2], [1, 1, 1])), end="")
    print(" | pairs(conv) %6.1f" % timeit(lambda: capi.rulebook_pairs(nb)), end="")
    print(" | row_order(nbrT) %6.1f us" % timeit(lambda: capi.rulebook_row_order(nbT)), flush=True)
    idx = oidx
