"""k_spconv_v9 (T tiles per wave, weight operands reused) against k_spconv_v6 on the encoder's level geometry:
bit-equality of the outputs and time per launch for every 32/64-channel layer shape (forward tables, transposed
tables with and without the mask row order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rslo_amd
from rslo_amd import capi, workload

FRAMES = int(os.environ.get("FRAMES", "8"))
REPS = int(os.environ.get("REPS", "30"))
MODES = [int(m) for m in os.environ.get("MODES", "0,2,4").split(",")]
torch.manual_seed(0)
net, _ = workload.build_network()
ex = workload.make_example(net, workload.kitti_pairs(FRAMES // 2))
coords = []
for t in range(2):
    c = ex["coordinates"][t].clone(); c[:, 0] += t * (FRAMES // 2); coords.append(c)
coords = torch.cat(coords, 0).contiguous()
idx = capi.SiteIndex(coords, FRAMES, [41, 768, 1408])
levels = [(idx, capi.rulebook_subm(idx, [3, 3, 3]))]
convs = []
for ks, st, pd in [([3,3,3],[2,2,2],[1,1,1]), ([3,3,3],[2,2,2],[1,1,1]), ([3,3,3],[2,2,2],[0,1,1])]:
    oidx, nbr, nbrT = capi.rulebook_conv(idx, ks, st, pd)
    convs.append((nbr, nbrT))
    levels.append((oidx, capi.rulebook_subm(oidx, [3, 3, 3])))
    idx = oidx
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
L = capi.lib()


ONLY = os.environ.get("ONLY", "")


def run(name, n_in, ci, co, nbr, order=None):
    if ONLY and ONLY not in name: return
    x, W, b = rn(n_in, ci), rn(27, ci, co) * 0.1, rn(co)
    Ws = capi.weight_split(W)
    P = int((nbr >= 0).sum())
    ref, line = None, "%-26s rows=%7d P=%8d" % (name, nbr.shape[0], P)
    for m in MODES:
        L.rslo_spconv_set_v9(m)
        y = capi.spconv_fwd_split(x, Ws, b, nbr, ci, co, act_slope=0.1, order=order); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS): capi.spconv_fwd_split(x, Ws, b, nbr, ci, co, act_slope=0.1, order=order)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / REPS
        if ref is None: ref = y
        same = bool(torch.equal(y, ref))
        line += "  | v9=%d %7.1f us %5.1f TF %s" % (m, us, 2 * P * ci * co / us / 1e6, "same" if same else "DIFF %.2e" % float((y - ref).abs().max()))
    L.rslo_spconv_set_v9(-1)
    print(line, flush=True)


for li, (ci, co) in [(1, (32, 32)), (2, (64, 64)), (3, (64, 64))]:
    nbr = levels[li][1]
    run("subm%d %d->%d" % (li, ci, co), nbr.shape[0], ci, co, nbr)
for ci_, ((nbr, nbrT), (a, b)) in enumerate(zip(convs, [(16, 32), (32, 64), (64, 64)])):
    if a >= 32:
        run("conv%d %d->%d" % (ci_, a, b), nbrT.shape[0], a, b, nbr)
        run("inv%d %d->%d" % (ci_, b, a), nbr.shape[0], b, a, nbrT)
        run("inv%d %d->%d mask order" % (ci_, b, a), nbr.shape[0], b, a, nbrT, order=capi.rulebook_row_order(nbrT))
# ragged / tiny sizes through v9
for n in (() if ONLY else (1, 31, 33, 257, 4097)):
    nb = levels[2][1][:n].clone(); nb[nb >= n] = -1
    run("subm2 64->64 first %d" % n, n, 64, 64, nb)
