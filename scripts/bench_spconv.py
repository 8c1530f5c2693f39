"""Micro-benchmark of the sparse-conv kernels on the encoder's real level geometry (8 synthetic frames):
per layer shape: time, algorithmic GB/s (SURVEY 8d formula), TFLOP/s, and a check against a torch reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rslo_amd
from rslo_amd import capi, synthetic, workload
import _tuning; _tuning.apply_from_env()

FRAMES = int(os.environ.get("FRAMES", "8"))
REPS = int(os.environ.get("REPS", "20"))
ONLY = os.environ.get("ONLY", "")
torch.manual_seed(0)
net, _ = workload.build_network()
clouds = workload.kitti_pairs(FRAMES // 2)
ex = workload.make_example(net, clouds)
coords = []
for t in range(2):
    c = ex["coordinates"][t].clone(); c[:, 0] += t * (FRAMES // 2); coords.append(c)
coords = torch.cat(coords, 0).contiguous()
idx0 = capi.SiteIndex(coords, FRAMES, [41, 768, 1408])
levels = [(idx0, capi.rulebook_subm(idx0, [3, 3, 3]))]
specs = [([3,3,3],[2,2,2],[1,1,1]), ([3,3,3],[2,2,2],[1,1,1]), ([3,3,3],[2,2,2],[0,1,1])]
convs = []
idx = idx0
for ks, st, pd in specs:
    oidx, nbr, nbrT = capi.rulebook_conv(idx, ks, st, pd)
    convs.append((nbr, nbrT))
    levels.append((oidx, capi.rulebook_subm(oidx, [3, 3, 3])))
    idx = oidx

def ref_conv(x, W, nbr):
    y = torch.zeros(nbr.shape[0], W.shape[2], device="cuda")
    for k in range(nbr.shape[1]):
        m = nbr[:, k] >= 0
        y[m] += x[nbr[m, k].long()] @ W[k]
    return y

ORDER = os.environ.get("ORDER", "none")     # tile order experiment: none | mask | morton | mortonN (block of 2^N cells) | random


def morton_key(c, shift=0):
    """(b, z, y, x) int32 -> int64 key: batch on top, then the bits of (z, y, x) >> shift interleaved, ties by row."""
    z, y, x = (c[:, 1].long() >> shift), (c[:, 2].long() >> shift), (c[:, 3].long() >> shift)
    key = torch.zeros_like(x)
    for b in range(11):
        key |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
    return (c[:, 0].long() << 40) | key


def make_order(nbr, coords):
    if ORDER == "none" or coords is None or coords.shape[0] != nbr.shape[0]:
        return None
    if ORDER == "mask":
        return capi.rulebook_row_order(nbr)
    if ORDER == "random":
        return torch.randperm(nbr.shape[0], device="cuda").int()
    shift = int(ORDER[6:] or 0)
    key = morton_key(coords, shift)
    return torch.sort(key, stable=True)[1].int()


LOCALIZE = os.environ.get("LOCALIZE", "0") == "1"   # upper bound of ANY row re-ordering: same masks (same MFMA work), but
# the neighbour of row o at offset k is row o + k - K/2: a tile's 27 neighbour sets are 27 overlapping contiguous runs


def localize(nbr):
    n, K = nbr.shape
    loc = (torch.arange(n, device=nbr.device, dtype=torch.int32)[:, None]
           + torch.arange(K, device=nbr.device, dtype=torch.int32)[None] - K // 2).clamp_(0, n - 1)
    return torch.where(nbr >= 0, loc, torch.full_like(nbr, -1)).contiguous()


def run(name, x, W, nbr, fn, check=True, coords=None):
    if ONLY and not any(o in name for o in ONLY.split(",")): return
    if LOCALIZE:
        nbr = localize(nbr)
    order = make_order(nbr, coords)
    P = int((nbr >= 0).sum()); n_out, K = nbr.shape; cin, cout = W.shape[1], W.shape[2]
    if fn == "dgrad":
        call = lambda: capi.spconv_dgrad(x, W, nbr, flip_k=True, order=order)
        cin_op, cout_op = cout, cin
    elif fn == "bf16":       # C4 feature path: bf16 rows in / out, fp32 accumulate
        xb = x.to(torch.bfloat16)
        call = lambda: capi.spconv_fwd_bf16(xb, W, None, nbr, order=order)
        cin_op, cout_op = cin, cout
    elif fn == "dgradT":     # dgrad as a forward conv with per-offset transposed weights (coalesced B reads)
        call = lambda: capi.spconv_fwd(x, W.transpose(1, 2).contiguous(), None, nbr, flip_k=True, order=order)
        cin_op, cout_op = cout, cin
    else:
        call = lambda: capi.spconv_fwd(x, W, None, nbr, order=order)
        cin_op, cout_op = cin, cout
    y = call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): call()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / REPS
    sz = 2 if fn == "bf16" else 4
    byts = P * cin_op * sz + n_out * cout_op * sz + 8 * P + K * cin * cout * sz
    fl = 2 * P * cin * cout
    err = ""
    if check and fn == "fwd":
        r = ref_conv(x, W, nbr); err = " maxrel=%.1e" % float((y - r).abs().max() / r.abs().max())
    print("%-28s rows=%7d P=%8d %8.1f us %7.1f GB/s %6.2f TF/s  (%.3f of 8TB/s)%s" % (name, n_out, P, us, byts / us / 1e3, fl / us / 1e6, byts / us / 1e3 / 8000, err), flush=True)

g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
chans = [(16, 16), (32, 32), (64, 64), (64, 64)]
for li, ((ix, nbr), (ci, co)) in enumerate(zip(levels, chans)):
    n = nbr.shape[0]
    run("subm%d %d->%d fwd" % (li, ci, co), rn(n, ci), rn(27, ci, co) * 0.1, nbr, "fwd", coords=ix.coords)
    run("subm%d %d->%d dgrad" % (li, ci, co), rn(n, co), rn(27, ci, co) * 0.1, nbr, "dgrad", check=False, coords=ix.coords)
    run("subm%d %d->%d dgradT" % (li, ci, co), rn(n, co), rn(27, ci, co) * 0.1, nbr, "dgradT", check=False, coords=ix.coords)
    if ci >= 32:
        run("subm%d %d->%d bf16" % (li, ci, co), rn(n, ci), rn(27, ci, co) * 0.1, nbr, "bf16", check=False, coords=ix.coords)
run("subm0 7->16 fwd", rn(levels[0][1].shape[0], 7), rn(27, 7, 16) * 0.1, levels[0][1], "fwd")
run("subm0 16->7 fwd", rn(levels[0][1].shape[0], 16), rn(27, 16, 7) * 0.1, levels[0][1], "fwd")
for ci, ((nbr, nbrT), (a, b)) in enumerate(zip(convs, [(16, 32), (32, 64), (64, 64)])):
    run("conv%d %d->%d fwd" % (ci, a, b), rn(nbrT.shape[0], a), rn(27, a, b) * 0.1, nbr, "fwd", coords=levels[ci + 1][0].coords)
    run("inv%d %d->%d fwd" % (ci, b, a), rn(nbr.shape[0], b), rn(27, b, a) * 0.1, nbrT, "fwd", coords=levels[ci][0].coords)
# wgrad
for li, ((ix, nbr), (ci, co)) in enumerate(zip(levels, chans)):
    if ONLY and "wgrad" not in ONLY: continue
    n = nbr.shape[0]; x = rn(n, ci); gy = rn(n, co)
    pairs = capi.rulebook_pairs(nbr)
    ref, _ = capi.spconv_wgrad(x, gy, nbr, ci, co)
    got, gb = capi.spconv_wgrad_pairs(x, gy, pairs, n, 27, ci, co); torch.cuda.synchronize()
    print("   wgrad pairs vs table maxrel=%.1e  bias maxrel=%.1e" % (float((got - ref).abs().max() / ref.abs().max()), float((gb - gy.sum(0)).abs().max() / gy.sum(0).abs().max())))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): capi.rulebook_pairs(nbr)
    e1.record(); torch.cuda.synchronize(); print("   rulebook_pairs %.1f us" % (1e3 * e0.elapsed_time(e1) / REPS))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS): capi.spconv_wgrad_pairs(x, gy, pairs, n, 27, ci, co)
    e1.record(); torch.cuda.synchronize()
    P = int((nbr >= 0).sum()); us = 1e3 * e0.elapsed_time(e1) / REPS
    byts = P * (ci + co) * 4 + 8 * P + 27 * ci * co * 4
    print("wgrad%d %d->%d                rows=%7d P=%8d %8.1f us %7.1f GB/s %6.2f TF/s" % (li, ci, co, n, P, us, byts / us / 1e3, 2 * P * ci * co / us / 1e6), flush=True)
