"""BASELINE configs C2 / C5 as encoder micro-benchmarks (parity-test configurations, not the bench line):
  c2: 1 x MI355X, forward-only GU encoder (+ covariance branch), one 64-ring synthetic scan, bs 1, fp32
  c5: 128-ring scans (~263 k points), 0.1 m cubic voxels (sparse shape [81,768,1408]), 4 frames, encoder fwd + bwd
Prints ms per pass and the algorithmic GB/s / TFLOP/s of the sparse convolutions (SURVEY.md 8d formula with the
measured pair counts).  Inputs are voxelized once and stay resident (the protocol of SURVEY.md 8d for C2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rslo_amd
from rslo_amd import capi, synthetic as S, workload
import spconv
from rslo.models import middle

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
torch.manual_seed(0)
if cfg == "c2":
    frames, n_el, vsize, max_vox, backward = 1, 64, S.VOXEL_SIZE, S.MAX_VOXELS, False
else:
    frames, n_el, vsize, max_vox, backward = 4, 128, S.VOXEL_SIZE_DENSE, 1 << 18, True
gen = spconv.utils.VoxelGenerator(list(vsize), list(S.PC_RANGE), S.MAX_POINTS_PER_VOXEL, max_vox)
grid = gen.grid_size
enc = middle.get_middle_class("SpMiddleFHDWithCov2_3")(
    [1] + grid[::-1].tolist() + [7], bn_type="None", use_leakyReLU=True, num_input_features=7, num_filters_down1=[],
    num_filters_down2=[]).cuda().train()
clouds = [torch.from_numpy(S.scan(n_el=n_el, scan_seed=i)).cuda() for i in range(frames)]
res = gen.generate_many(clouds, max_vox)
feats = torch.cat([capi.vfe_mean(v, n) for v, c, n in res], 0)
coords = torch.cat([torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32, device="cuda"), c], 1)
                    for b, (v, c, n) in enumerate(res)], 0)
print("config %s: %d frame(s), %d points/frame, %d voxels, sparse shape %s" %
      (cfg, frames, clouds[0].shape[0], feats.shape[0], list(enc.sparse_shape)), flush=True)

# algorithmic bytes / flops of the sparse convs from the measured pair counts
byts = fl = 0
plan = enc.plan(coords, frames)
convs = [m for seq in (enc.middle_conv, enc.middle_conv_tail, enc.middle_cov_deconv) for m in seq if isinstance(m, spconv.SparseConvolution)]
for m in convs:
    rb = plan.indice_dict[m.indice_key]
    t = rb.nbrT if m.inverse else rb.nbr
    P = int((t >= 0).sum()); n_out, K = t.shape
    byts += P * m.in_channels * 4 + n_out * m.out_channels * 4 + 8 * P + K * m.in_channels * m.out_channels * 4
    fl += 2 * P * m.in_channels * m.out_channels

def once():
    x = feats.clone().requires_grad_(backward)
    bev, cov = enc(x, coords, frames)
    if backward:
        (bev.square().mean() + cov.square().mean()).backward()
        enc.zero_grad(set_to_none=True)
for _ in range(5): once()
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for _ in range(N): once()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / N * 1e3
mult = 3.0 if backward else 1.0
report = {"config": cfg, "frames": frames, "points_per_frame": int(clouds[0].shape[0]), "voxels": int(feats.shape[0]),
          "sparse_shape": [int(v) for v in enc.sparse_shape], "pass": "fwd+bwd" if backward else "forward", "sparse_convs": len(convs),
          "ms_per_pass_incl_rulebooks": round(ms, 3), "algorithmic_GB": round(mult * byts / 1e9, 3),
          "algorithmic_GFLOP": round(mult * fl / 1e9, 2), "algorithmic_GBps": round(mult * byts / ms / 1e6, 1),
          "hbm_roofline_frac": round(mult * byts / ms / 1e6 / 8000.0, 4),
          "ideal_ms_at_8TBps": round(mult * byts / 8e9, 4)}
# the same pass on rulebooks planned once (a data loader can build them ahead: they depend on coordinates only)
def planned():
    x = feats.clone().requires_grad_(backward)
    bev, cov = enc(x, coords, frames, plan=plan)
    if backward:
        (bev.square().mean() + cov.square().mean()).backward()
        enc.zero_grad(set_to_none=True)
    return bev
for _ in range(5): planned()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): planned()
torch.cuda.synchronize(); ms_p = (time.perf_counter() - t0) / N * 1e3
report["ms_per_pass_planned"] = round(ms_p, 3)
report["hbm_roofline_frac_planned"] = round(mult * byts / ms_p / 1e6 / 8000.0, 4)
if not backward:
    # launch-bound single-frame forward: replay the ~30 launches from one hipGraph
    try:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3): planned()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_static = planned()
        for _ in range(5): graph.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N): graph.replay()
        torch.cuda.synchronize(); ms_g = (time.perf_counter() - t0) / N * 1e3
        ref = planned()
        report["ms_per_pass_planned_hipgraph"] = round(ms_g, 3)
        report["hipgraph_equals_eager"] = bool(torch.equal(out_static, ref))
        report["hbm_roofline_frac_hipgraph"] = round(mult * byts / ms_g / 1e6 / 8000.0, 4)
    except Exception as e:      # recorded, not fatal: the eager numbers above stand
        report["hipgraph_error"] = repr(e)[:300]
if len(sys.argv) > 2:
    import json
    json.dump(report, open(sys.argv[2], "w"), indent=1)
print(report)
print("%s: %.2f ms per %s pass (rulebooks + %d sparse convs%s); sparse-conv algorithmic %.2f GB, %.1f GFLOP per pass -> %.0f GB/s, %.1f TFLOP/s over the whole pass"
      % (cfg, ms, "fwd+bwd" if backward else "forward", len(convs), " + dense()", mult * byts / 1e9, mult * fl / 1e9,
         mult * byts / ms / 1e6, mult * fl / ms / 1e9))
