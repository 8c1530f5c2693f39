#!/usr/bin/env python
"""bench.py -- frame-pairs/s of the RSLO two-frame odometry hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N = 1: single process)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, RCCL)

Workload (BASELINE.json configs[2], "C3"): full forward + backward of UnVoxelOdomNetICP3 on synthetic
KITTI-shaped frame pairs (2 x ~131k points each), batch 4 frame pairs per GPU, fp32, random-init
weights of the shipped architecture.  One step = voxelize the 8 resident point clouds -> VFE -> batched sparse
GU encoder (+ covariance branch) -> BEV head + ego-motion vote -> consistency loss (chamfer NN, covariance
residual, ICP) -> backward -> Adam step.  Raw points are resident in HBM before the timed region.
N > 1: pure data parallel (each rank its own frame pairs, weak scaling), DDP gradient all-reduce over RCCL.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant hand-written kernel (the sparse-conv
implicit GEMM): algorithmic bytes (SURVEY.md 8d formula) / HIP-event time per launch, measured live in the
timed region.  `cpu_baseline` is the same path on the host cores over the oracle backend, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# The step keeps five HIP streams busy (training, structure plan, covariance branch, weight gradients, RCCL): with the
# runtime's default of 4 hardware queues two of them share a queue and serialise -- a sixth stream put the training
# stream behind a side stream for a 23 ms step.  Must be set before the HIP runtime starts.
if os.environ.get("RSLO_BENCH_ONE_GPU", "0") != "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
else:      # ranks sharing ONE GPU (functional mode): their queues must all stay mapped -- a kernel of the peer exchange spins
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")      # until the other process's kernel runs (unset: 279 ms per step; 2: 40)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E ~8 TB/s (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3   # dense fp32-input MFMA peak = the fp32 vector peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline includes 2:1 sparsity)


def lib_hash():
    """rslo_amd.build.source_hash(): first 16 hex digits of the sha256 over the sources and flags librslo_hip.so is built from.
    The PMC summaries under profiles/ carry the hash of the library they were collected with; counter fields are only quoted
    when it matches the library being benchmarked."""
    try:
        import runpy
        return runpy.run_path(os.path.join(ROOT, "rslo_amd", "build.py"))["source_hash"]()
    except OSError:
        return None


def _head_graph_mode(net):
    from rslo_amd import headgraph
    st = headgraph._STATE.get(getattr(net, "odom_predictor", None))
    return st.graph.mode if (st is not None and st.graph is not None) else None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="frame pairs per GPU (C3: 4)")
    ap.add_argument("--rings", type=int, default=64, help="64 = KITTI-shaped ~131k points; 128 = dense scan")
    ap.add_argument("--no-optim", action="store_true", help="time forward+backward only")
    ap.add_argument("--no-prefetch", action="store_true", help="voxelize / plan inside the step on the training stream")
    ap.add_argument("--no-voxelize", action="store_true", help="voxelize once outside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (roofline=null)")
    ap.add_argument("--kernel-report", default="", help="write the per-kernel roofline table to this JSON file")
    ap.add_argument("--config", default=None, choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs[1..4]: c2 = forward-only GU encoder, one 64-ring scan, bs 1, fp32; c3 = the "
                         "headline line (default): full fwd+bwd, bs 4, fp32; c4 = the same step with bf16 conv operands / "
                         "trunk features (per-GPU part at N = 1, the full config with --gpus 8); c5 = 128-ring scans at 0.1 m "
                         "voxels, bs 2 per GPU, GU encoder fwd+bwd (its 256 BEV channels do not fit the head: SURVEY 8d)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / timing protocol only, over gloo on the CPU with an empty step: what the "
                         "non-GPU test of `--gpus N` runs (no kernels, value is not a measurement)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = C3 (the headline line); bf16 = the single-GPU half of C4: apex.amp O1, bf16 conv operands")
    args = ap.parse_args()
    if args.config is None:
        args.config = "c4" if args.dtype == "bf16" else "c3"
    if args.config == "c4":
        args.dtype = "bf16"
    if args.config == "c5" and args.batch == 4 and "--batch" not in " ".join(sys.argv):
        args.batch = 2                      # C5: bs 2 frame pairs per GPU
    if args.config == "c2":
        args.batch = 1
    return args


def pin_to_quiet_cores(n=8, part=(0, 1)):
    """The step is bound by how fast two Python threads issue ~1000 launches; on the shared boxes of the pool the
    scheduler migrates them between busy cores (same binary, same box: 14.8 ... 17.4 ms per step).  Pick the block of n
    consecutive allowed CPUs that was idlest over a 100 ms sample of /proc/stat and keep the process there
    (RSLO_BENCH_PIN=0 turns it off; measured on one box, alternating runs: 14.68-14.77 ms per step pinned against
    14.9 / 15.9 / 16.6 / 18.4 ms free, host issue 5.1 vs 7-8.6 ms).  part = (local rank, ranks on the node): rank r only
    considers every ranks-th block, so the ranks of a node never share cores.  Returns the chosen CPUs or None."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if len(allowed) <= n:
            return None

        def sample():
            out = {}
            with open("/proc/stat") as f:
                for line in f:
                    if line.startswith("cpu") and line[3].isdigit():
                        p = line.split()
                        v = [int(x) for x in p[1:9]]
                        out[int(p[0][3:])] = (sum(v), v[3] + v[4])      # total, idle + iowait
            return out
        a = sample()
        time.sleep(0.1)
        b = sample()
        busy = {c: 1.0 - (b[c][1] - a[c][1]) / max(b[c][0] - a[c][0], 1) for c in allowed if c in a and c in b}
        best, best_load = None, None
        for i in range(0, len(allowed) - n + 1):
            blk = allowed[i:i + n]
            if blk[-1] - blk[0] != n - 1 or blk[0] % n:      # aligned blocks of consecutive ids: one core complex
                continue
            if (blk[0] // n) % max(part[1], 1) != part[0] % max(part[1], 1):
                continue
            load = sum(busy.get(c, 1.0) for c in blk)
            if best is None or load < best_load:
                best, best_load = blk, load
        if best is None:
            return None
        os.sched_setaffinity(0, set(best))
        return best
    except (AttributeError, OSError, ValueError):
        return None


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


PMC_TRAFFIC = "r06_pmc_traffic_bench.json"
PMC_BUSY = "r06_pmc_mfma_busy.json"


# --------------------------------------------------------------------------------------------- kernel events
class ConvProbe:
    """Wraps the sparse-conv C-ABI calls with HIP events on the launching stream and records the
    algorithmic bytes / flops of every launch (P*Cin*s + Nout*Cout*s + 8P + K*Cin*Cout*s)."""

    def __init__(self, capi):
        self.capi = capi
        self.records = []
        self.enabled = False
        self.keep_tables = False
        self.orders = {}          # table -> tile order of the launch (k_spconv_v6 walks 32-row tiles of the ORDERED rows)
        self._orig = {}

    @staticmethod
    def kernel_name(cin_op, cout_op, n_out, trans, split=False, bf16=False):
        """The template instantiation the entry point dispatches to (rslo_amd/csrc/spconv.hip)."""
        t = "true" if trans else "false"
        rbw = 2 if n_out >= 256 * 32 * 8 else 1
        if bf16:
            return "k_spconv_bf16<%d, %d, %d, %d>" % (cin_op, cout_op, rbw, 1 if (cin_op == 32 and cout_op == 32) else 2)
        if split:      # rslo_spconv_fwd_split: two waves per 32-row tile except for 32 -> 32; dead 16-row blocks skipped
            ks = 1 if (cin_op == 32 and cout_op == 32) else 2
            if n_out < 20000:                                      # the small-problem tiling: 16-row tiles, four waves
                return "k_spconv_v6<%d, %d, 1, 4, false>" % (cin_op, cout_op)
            rb = 2 if ks == 2 else rbw
            if rb == 1:
                return "k_spconv_v6<%d, %d, 1, 4, false>" % (cin_op, cout_op)
            return "k_spconv_v6<%d, %d, 2, %d, true>" % (cin_op, cout_op, ks)
        if cin_op % 16 == 0 and cout_op % 16 == 0:
            return "k_spconv_v3<%d, %d, %d, %s>" % (cin_op, cout_op, rbw, t)
        ci = 8 if cin_op <= 8 else (16 if cin_op <= 16 else (32 if cin_op <= 32 else 64))
        co = 16 if cout_op <= 16 else (32 if cout_op <= 32 else 64)
        rb = 2 if (n_out >= 256 * 128 * 2 and co >= 32) else 1
        return "k_spconv<%d, %d, %d, %s>" % (ci, co, rb, t)

    def install(self):
        capi = self.capi
        # the lowest-level wrappers: capi.spconv_fwd / spconv_dgrad only dispatch to these (the data gradient of
        # MFMA-shaped layers runs through a forward kernel on transposed weights)
        names = ("spconv_fwd_direct", "spconv_fwd_split", "spconv_dgrad_direct", "spconv_fwd_bf16")
        self._orig = {n: getattr(capi, n) for n in names}
        probe = self

        def timed(fn, name_fn):
            def wrapper(*a, **k):
                if not probe.enabled or torch.cuda.is_current_stream_capturing():      # (rslo_amd/headgraph.py capturing the head)
                    return fn(*a, **k)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **k)
                e1.record()
                probe.records.append((name_fn(*a, **k), e0, e1))
                return out
            return wrapper

        def fwd_meta(x, W, bias, nbr, flip_k=False, act_slope=1.0, order=None, n_live=None):
            K, cin, cout = W.shape
            return ("fwd", cin, cout, K, nbr.shape[0], nbr if probe.keep_tables else None,
                    probe.kernel_name(cin, cout, nbr.shape[0], False))

        def split_meta(x, Ws, bias, nbr, cin, cout, flip_k=False, act_slope=1.0, order=None, n_live=None):
            if probe.keep_tables and order is not None:
                probe.orders[(nbr.data_ptr(), tuple(nbr.shape))] = order
            return ("fwd", cin, cout, nbr.shape[1], nbr.shape[0], nbr if probe.keep_tables else None,
                    probe.kernel_name(cin, cout, nbr.shape[0], False, split=True))

        def bf16_meta(x, W, bias, nbr, flip_k=False, act_slope=1.0, transpose=False, order=None):
            K, cin, cout = W.shape
            ci, co = (cout, cin) if transpose else (cin, cout)
            return ("bf16", ci, co, K, nbr.shape[0], nbr if probe.keep_tables else None,
                    probe.kernel_name(ci, co, nbr.shape[0], False, bf16=True))

        def dgrad_meta(dout, W, nbrT, flip_k=False, order=None):
            K, cin, cout = W.shape
            return ("dgrad", cout, cin, K, nbrT.shape[0], nbrT if probe.keep_tables else None,
                    probe.kernel_name(cout, cin, nbrT.shape[0], True))

        # dense 3x3 convolutions of the BEV head (csrc/conv2d.hip): algorithmic flops 2 B Ho Wo Cin Cout 9; bytes = input +
        # output (+ split weights / + weight gradient), each touched once
        def c2f_meta(x, ws, bias, cout, lp=False, residual=None):
            B, cin, H, W = x.shape
            # the instantiation conv2d_fwd_launch picks (csrc/conv2d.hip): second wave set on long chains / empty CUs
            wgs = B * ((W + 15) // 16) * ((H + 3) // 4) * (cout // 32)
            kc = 2 if (cin >= 64 and cin >= 256 and wgs <= 200) else 1
            res_bytes = 4 * B * H * W * cout if residual is not None else 0      # the epilogue's residual read
            if not lp and cin <= 64 and cout >= 192 and cout % 64 == 0 and H * W >= 96 * 176:
                inst = "k_conv2d_fwd<4, 2, false, false, 1, 2>"      # two channel blocks per wave (conv2d_fwd_plan)
            elif not lp and kc == 1 and wgs >= 512:      # >= 2 workgroups per CU: the 96-register one-tap-ahead variant
                inst = "k_conv2d_fwd<4, 1, false, false, 1, 5>"
            else:
                inst = "k_conv2d_fwd<4, 1, true, %s, %d, %d>" % ("true" if lp else "false", kc, 2 // kc)
            return ("dense", 2 * B * H * W * cin * cout * 9, 4 * B * H * W * (cin + cout) + 54 * cin * cout + res_bytes,
                    "%s [%d->%d %dx%d]" % (inst, cin, cout, H, W))

        def c2w_meta(x, dout, stride=1, want_bias=False, lp=False):
            B, cin, H, W = x.shape
            cout, Ho, Wo = dout.shape[1], dout.shape[2], dout.shape[3]
            return ("dense", 2 * B * Ho * Wo * cin * cout * 9, 4 * B * (H * W * cin + Ho * Wo * cout) + 36 * cin * cout,
                    ("k_conv2d_wgrad_s1<2, %s>" % ("true" if lp else "false") if stride == 1 else "k_conv2d_wgrad<2, 2>")
                    + " [%d->%d %dx%d]" % (cin, cout, H, W))

        for n, meta in (("conv2d_fwd", c2f_meta), ("conv2d_wgrad", c2w_meta)):
            self._orig[n] = getattr(capi, n)
            setattr(capi, n, timed(self._orig[n], meta))
        capi.spconv_fwd_direct = timed(self._orig["spconv_fwd_direct"], fwd_meta)
        capi.spconv_fwd_split = timed(self._orig["spconv_fwd_split"], split_meta)
        capi.spconv_dgrad_direct = timed(self._orig["spconv_dgrad_direct"], dgrad_meta)
        capi.spconv_fwd_bf16 = timed(self._orig["spconv_fwd_bf16"], bf16_meta)

    def uninstall(self):
        for k, f in self._orig.items():
            setattr(self.capi, k, f)

    def summarize(self, steps):
        """-> (per-kernel dict, dominant kernel roofline dict).  The probed steps are recorded with keep_tables so that
        every launch's own pair count is read (the tables stay alive until this summary)."""
        torch.cuda.synchronize()
        counted = {}      # pair count per table (the probed steps run DIFFERENT batches: every record has its own table)

        def n_pairs(t):
            k = (t.data_ptr(), tuple(t.shape))
            if k not in counted:
                counted[k] = int((t >= 0).sum().item())
            return counted[k]
        issued_cache = {}

        def issued_pairs(t, rows_per_tile=32):
            """Row-offset products the tiled kernel ISSUES for this table: every offset that is active for any row of a
            32-row tile is computed for all 32 rows (padding = the part of the offset union a row does not have)."""
            k = (t.data_ptr(), tuple(t.shape))
            ck = k + (rows_per_tile,)
            if ck not in issued_cache:
                a = t >= 0
                o = self.orders.get(k)
                if o is not None and o.numel() == a.shape[0]:
                    a = a[o.long().clamp_(0, a.shape[0] - 1)]      # (a recycled arena could hold anything: never fault)
                n, K = a.shape
                pad = (-n) % rows_per_tile
                if pad:
                    a = torch.cat([a, torch.zeros((pad, K), dtype=torch.bool, device=a.device)], 0)
                issued_cache[ck] = int(a.view(-1, rows_per_tile, K).any(dim=1).sum().item()) * rows_per_tile
            return issued_cache[ck]
        groups = {}
        for i, (m, e0, e1) in enumerate(self.records):
            if m[0] == "dense":
                _, flops, byts, name = m
            else:
                kind, cin, cout, K, n_out, table, name = m
                P = n_pairs(table)
                sz = 2 if kind == "bf16" else 4          # bytes per feature / weight element
                byts = P * cin * sz + n_out * cout * sz + 8 * P + K * cin * cout * sz
                flops = 2 * P * cin * cout
            g = groups.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0, "pairs": 0, "issued_pairs": 0,
                                         "big": {"launches": 0, "ms": 0.0, "flops": 0}})
            if m[0] != "dense" and "k_spconv_v6" in name:
                g["pairs"] += P
                # the skipping kernels (k_spconv_v6<.., true>) issue products per live 16-row BLOCK of a (tile, offset)
                g["issued_pairs"] += issued_pairs(table, 16 if name.endswith("true>") else 32)
            dt = e0.elapsed_time(e1)
            g["launches"] += 1
            g["ms"] += dt
            g["bytes"] += byts
            g["flops"] += flops
            if m[0] != "dense" and n_out >= 256 * 32 * 8:      # the launches the round-1 / round-2 lines called <.., 2>
                g["big"]["launches"] += 1
                g["big"]["ms"] += dt
                g["big"]["flops"] += flops
        for g in groups.values():
            g["avg_us"] = 1e3 * g["ms"] / g["launches"]
            g["GBps"] = g["bytes"] / (g["ms"] * 1e-3) / 1e9
            g["TFLOPs"] = g["flops"] / (g["ms"] * 1e-3) / 1e12
        # PMC summaries (scripts/pmc_bench_traffic.sh, scripts/pmc_mfma_busy.sh) are quoted only when they were collected
        # with the library being benchmarked (they record its hash); otherwise the fields stay null
        here = lib_hash()

        def load_pmc(fname):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    d = json.load(f)
            except (OSError, ValueError):
                return {}, "profiles/%s missing" % fname
            if d.get("lib_sha256") != here:
                return {}, "profiles/%s was collected with another build of librslo_hip.so (%s, now %s)" % (
                    fname, d.get("lib_sha256"), here)
            return d.get("kernels", {}), None
        if getattr(self, "pmc_workload_ok", True):
            pmc, pmc_note = load_pmc(PMC_TRAFFIC)
            pmc_busy, busy_note = load_pmc(PMC_BUSY)
        else:      # the PMC passes were collected on the C3 step: per-launch figures of other workloads are not theirs
            pmc, pmc_busy = {}, {}
            pmc_note = busy_note = "PMC summaries under profiles/ are of the C3 workload (scripts/pmc_*.sh)"

        def roof_of(gname):
            g = groups[gname]
            name = gname.split(" [")[0]          # dense groups are keyed kernel + layer shape
            # which roof bounds it: arithmetic intensity vs the fp32 ridge (157.3 TF / 8 TB/s = 19.7 flop/B)
            ai = g["flops"] / g["bytes"]
            lowp = ("k_spconv_bf16" in name or (name.startswith("k_conv2d_wgrad_s1") and name.endswith(", true>"))
                    or (name.startswith("k_conv2d_fwd") and name.split(", ")[3] == "true"))      # bf16 operands (C4)
            if lowp and ai >= MFMA_BF16_PEAK_TF * 1e3 / HBM_PEAK_GBS:
                r = {"bound": "mfma", "kernel": name, "achieved": round(g["TFLOPs"], 2), "peak": MFMA_BF16_PEAK_TF,
                     "unit": "TFLOP/s", "frac": round(g["TFLOPs"] / MFMA_BF16_PEAK_TF, 4), "traffic": None}
            elif not lowp and (ai >= (MFMA_BF16_PEAK_TF / 6.0) * 1e3 / HBM_PEAK_GBS or "k_spconv" not in name):
                # fp32-path kernels issue six bf16 MFMAs per product block: the matrix cores can sustain 2500 / 6 = 417
                # TFLOP/s of fp32-equivalent work for THAT instruction stream, so the ridge is 417 TF / 8 TB/s = 52 flop/B.
                # The sparse gather kernels (28.8 flop/B) are below it: HBM is their binding roof (SURVEY.md 8d and the
                # north star name it); the dense convolutions (~200 flop/B) are matrix-core bound.
                r = {"bound": "mfma", "kernel": name, "achieved": round(g["TFLOPs"], 2), "peak": MFMA_F32_PEAK_TF,
                     "unit": "TFLOP/s", "frac": round(g["TFLOPs"] / MFMA_F32_PEAK_TF, 4), "traffic": None}
                # the kernels issue six bf16 products per fp32 product: what the matrix cores could deliver for THIS
                # instruction stream is 2500 / 6 TFLOP/s of fp32-equivalent work
                r["frac_of_bf16_peak_over_6"] = round(6 * g["TFLOPs"] / MFMA_BF16_PEAK_TF, 4)
            else:
                r = {"bound": "hbm", "kernel": name, "achieved": round(g["GBps"], 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(g["GBps"] / HBM_PEAK_GBS, 4), "traffic": None}
            if g.get("issued_pairs"):
                # useful / issued MFMA work of the tiled gather: a tile computes its offset UNION for all of its 32 rows
                r["mfma_useful_frac"] = round(g["pairs"] / g["issued_pairs"], 4)
            if lowp:
                r["matrix_core_path"] = "bf16 operands (round to nearest), fp32 accumulation: 1 x v_mfma_f32_16x16x32_bf16"
            elif "k_spconv_v6" in name or "k_conv2d" in name:
                # fp32 products computed as six bf16 MFMAs on exactly split operands (csrc/spconv.hip v6, csrc/conv2d.hip):
                # the algorithmic fp32 flops are priced against the fp32 MFMA peak; the bf16 matrix-core work issued is 6x
                r["matrix_core_path"] = "fp32 = 3-way exact bf16 split, 6 x v_mfma_f32_16x16x32_bf16 per product block"
                r["issued_bf16_TFLOPs"] = round(6 * g["TFLOPs"], 1)
                r["issued_bf16_frac_of_2500TF"] = round(6 * g["TFLOPs"] / 2500.0, 4)
            r.update({"arithmetic_intensity_flop_per_byte": round(ai, 2), "algorithmic_GBps": round(g["GBps"], 1),
                      "hbm_frac": round(g["GBps"] / HBM_PEAK_GBS, 4), "algorithmic_TFLOPs": round(g["TFLOPs"], 2),
                      "avg_launch_us": round(g["avg_us"], 2), "launches_per_step": g["launches"] // max(steps, 1),
                      "ms_per_step": round(g["ms"] / max(steps, 1), 3),
                      "algorithmic_bytes_per_launch": int(g["bytes"] // g["launches"]),
                      "algorithmic_flops_per_launch": int(g["flops"] // g["launches"])})
            # HBM-side bytes per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md), collected
            # with rocprofv3 on this same command (scripts/pmc_bench_traffic.sh) and committed under profiles/
            if gname != name:
                r["layer_shape"] = gname.split(" [")[1].rstrip("]")
            big = g.get("big")
            if big and 0 < big["launches"] < g["launches"] and r["bound"] == "mfma":
                # one kernel name now serves every 64-channel layer; the subset with >= 65536 output rows is the set of
                # launches earlier lines reported as k_spconv_v6<64, 64, 2> (comparable round over round)
                tf = big["flops"] / (big["ms"] * 1e-3) / 1e12
                r["launches_with_65536_rows_or_more"] = {
                    "launches_per_step": big["launches"] // max(steps, 1),
                    "avg_launch_us": round(1e3 * big["ms"] / big["launches"], 2), "achieved": round(tf, 2),
                    "frac": round(tf / r["peak"], 4)}
            if name in pmc_busy:       # matrix-core busy cycles / (shader cycles x 1024 SIMDs), scripts/pmc_mfma_busy.sh
                r["mfma_busy_pct"] = pmc_busy[name]["mfma_busy_pct"]
                r["mfma_busy_source"] = "profiles/" + PMC_BUSY
            elif busy_note:
                r["mfma_busy_note"] = busy_note
            if name in pmc:
                r["traffic"] = int(pmc[name]["hbm_bytes_per_launch"])
                r["traffic_source"] = "profiles/" + PMC_TRAFFIC + (
                    " (average over all layer shapes of this kernel)" if gname != name else "")
            elif pmc_note:
                r["traffic_note"] = pmc_note
            return r

        # The dominant kernel = the kernel INSTANTIATION with the largest total time per step over ALL probed families
        # (a dense kernel serves many layer shapes: its launches are summed; per-shape entries stay in `groups`).  The
        # sparse gather kernel the north star names is reported next to it on the HBM roofline.
        merged = {}
        for gname, g in groups.items():
            k = gname.split(" [")[0]
            m = merged.setdefault(k, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0, "pairs": 0, "issued_pairs": 0, "big": None})
            for f in ("launches", "ms", "bytes", "flops", "pairs", "issued_pairs"):
                m[f] += g.get(f, 0)
            if gname == k:
                m["big"] = g.get("big")
        for k, m in merged.items():
            m["avg_us"] = 1e3 * m["ms"] / m["launches"]
            m["GBps"] = m["bytes"] / (m["ms"] * 1e-3) / 1e9
            m["TFLOPs"] = m["flops"] / (m["ms"] * 1e-3) / 1e12
            if k not in groups:
                groups[k] = m                    # the all-shapes entry of a dense kernel
        by_ms = sorted(merged, key=lambda n: -merged[n]["ms"])
        # The dense weight gradients run as leaf work on their own HIP stream (rslo_amd/streams.py), beside the training
        # stream's launches: events around them measure that shared time (61 us against 31 us in the rocprofv3 trace of the
        # same command), and they are off the step's critical path.  The dominant kernel is taken over the training
        # stream's kernels; the weight-gradient group stays in the report as `leaf_stream_kernel`.
        from rslo_amd import streams as _streams
        leaf = [n for n in by_ms if n.startswith("k_conv2d_wgrad")] if _streams.ENABLED else []
        ranked = [n for n in by_ms if n not in leaf] or by_ms
        top = ranked[0]
        roof = roof_of(top)
        if "k_conv2d" in top:
            shapes = sorted((n for n in groups if n.startswith(top + " [")), key=lambda n: -groups[n]["ms"])
            roof["layer_shapes"] = [{"shape": n.split(" [")[1].rstrip("]"), "launches_per_step": groups[n]["launches"] // max(steps, 1),
                                     "avg_launch_us": round(groups[n]["avg_us"], 2), "algorithmic_TFLOPs": round(groups[n]["TFLOPs"], 2)}
                                    for n in shapes]
        sparse = [n for n in by_ms if "k_conv2d" not in n]
        if sparse:
            sp = roof_of(sparse[0])
            # the gather kernel against the roofline SURVEY.md 8d / the north star name for it: algorithmic bytes per
            # launch / launch time over 8 TB/s (its fp32-MFMA view -- `achieved` / `frac` above -- stays in the object)
            sp["hbm_roofline"] = {"bound": "hbm", "achieved": sp["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": sp["hbm_frac"], "traffic": sp.get("traffic")}
            roof["sparse_gather_kernel"] = sp
        dense = [n for n in ranked if "k_conv2d" in n]
        if dense and "k_conv2d" not in top:
            roof["dense_top_kernel"] = roof_of(dense[0])
        if leaf:
            roof["leaf_stream_kernel"] = roof_of(leaf[0])
            roof["leaf_stream_kernel"]["note"] = ("runs on the weight-gradient stream beside the training stream: the event "
                                                  "time includes the sharing; rocprofv3 duration in profiles/*_kernel_stats.csv")
        tot_ms = sum(x["ms"] for n, x in merged.items() if "k_conv2d" not in n)
        tot_b = sum(x["bytes"] for n, x in merged.items() if "k_conv2d" not in n)
        roof["all_spconv_fwd_dgrad"] = {"ms_per_step": round(tot_ms / max(steps, 1), 3),
                                        "GBps": round(tot_b / (tot_ms * 1e-3) / 1e9, 1)}
        tot_d = sum(x["ms"] for n, x in merged.items() if "k_conv2d" in n)
        tot_f = sum(x["flops"] for n, x in merged.items() if "k_conv2d" in n)
        if tot_d > 0:
            roof["all_dense_conv_probed"] = {"ms_per_step": round(tot_d / max(steps, 1), 3),
                                             "TFLOPs": round(tot_f / (tot_d * 1e-3) / 1e12, 1)}
        return groups, roof


# --------------------------------------------------------------------------------------------- cpu baseline
def cpu_baseline(args):
    """The same host modules over the oracle backend on the host cores: 1 frame pair, forward + backward
    (bounded sample: ~10-30 s of CPU work)."""
    from oracle import cpu_backend
    from rslo_amd import workload
    torch.manual_seed(7)
    threads = torch.get_num_threads()
    with cpu_backend.patched():
        net, _ = workload.build_network(device="cpu")
        net.train()
        net.global_step.fill_(2000)
        clouds = workload.kitti_pairs(1, n_el=args.rings)
        t0 = time.time()
        ex = workload.make_example(net, clouds, device="cpu")
        ret = net(ex)
        ret["loss"].backward()
        dt = time.time() - t0
    return {"value": round(1.0 / dt, 4), "unit": "frame-pairs/s", "cores": int(threads), "cpu_model": cpu_model(),
            "kind": "port",
            "sample": "1 frame pair (2 x %d-ring synthetic scans), voxelize+forward+backward, bs=1, "
                      "oracle C/OpenMP sparse ops + torch-CPU dense head; %.1f s" % (args.rings, dt)}


# --------------------------------------------------------------------------------------------- C2 / C5: GU encoder
def encoder_setup(cfg, device, frames, seed0=0):
    """The GU encoder (SpMiddleFHDWithCov2_3 incl. covariance branch) of the shipped config with resident, voxelized
    inputs (C2 times the encoder on inputs already voxelized on the device).  C5: 128-ring scans, 0.1 m
    cubic voxels, max_voxels lifted to 2^18."""
    import spconv
    from rslo.models import middle
    from rslo_amd import synthetic as S
    dev = torch.device(device)
    if cfg == "c2":
        n_el, vsize, max_vox = 64, S.VOXEL_SIZE, S.MAX_VOXELS
    else:
        n_el, vsize, max_vox = 128, S.VOXEL_SIZE_DENSE, 1 << 18
    gen = spconv.utils.VoxelGenerator(list(vsize), list(S.PC_RANGE), S.MAX_POINTS_PER_VOXEL, max_vox)
    grid = gen.grid_size
    torch.manual_seed(7)
    enc = middle.get_middle_class("SpMiddleFHDWithCov2_3")(
        [1] + grid[::-1].tolist() + [7], bn_type="None", use_leakyReLU=True, num_input_features=7, num_filters_down1=[],
        num_filters_down2=[]).to(dev).train()
    clouds = [S.scan(n_el=n_el, scan_seed=seed0 + i) for i in range(frames)]
    return enc, gen, clouds, max_vox


def encoder_inputs(gen, clouds, max_vox, dev):
    from rslo_amd import capi
    if dev.type == "cuda":
        res = gen.generate_many([torch.from_numpy(c).to(dev) for c in clouds], max_vox)
    else:
        res = []
        for c in clouds:
            r = gen.generate(c, max_vox)
            if isinstance(r, dict):
                r = tuple(r[k] for k in ("voxels", "coordinates", "num_points_per_voxel"))
            res.append(tuple(torch.as_tensor(a) for a in r[:3]))
    feats = torch.cat([capi.vfe_mean(v, n) for v, c, n in res], 0)
    coords = torch.cat([torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32, device=c.device), c], 1)
                        for b, (v, c, n) in enumerate(res)], 0)
    return feats, coords


def encoder_cpu_baseline(args):
    """One frame of the same encoder pass over the oracle backend on the host cores."""
    from oracle import cpu_backend
    cfg = args.config
    backward = cfg == "c5"
    threads = torch.get_num_threads()
    with cpu_backend.patched():
        enc, gen, clouds, max_vox = encoder_setup(cfg, "cpu", 1)
        feats, coords = encoder_inputs(gen, clouds, max_vox, torch.device("cpu"))
        t0 = time.time()
        n = 0
        while n < 12 and (n == 0 or time.time() - t0 < 10.0):       # a bounded sample: >= 10 s or 12 passes
            x = feats.clone().requires_grad_(backward)
            bev, cov = enc(x, coords, 1)
            if backward:
                (bev.square().mean() + cov.square().mean()).backward()
                enc.zero_grad(set_to_none=True)
            n += 1
        dt = (time.time() - t0) / n
    return {"value": round(0.5 / dt, 4), "unit": "frame-pairs/s", "cores": int(threads), "cpu_model": cpu_model(),
            "kind": "port",
            "sample": "%d passes over 1 frame (%d voxels) = half a frame pair each, rulebooks + 20 sparse convs + dense(), %s, "
                      "oracle C/OpenMP; %.2f s per pass" % (n, feats.shape[0], "forward + backward" if backward else "forward", dt)}


def run_encoder(args, world, rank, local_rank, dist_on, dev):
    """C2 / C5 (BASELINE.json configs[1], configs[4]) through the same entry point and JSON schema as the C3 line.
    A step = one pass of the GU encoder over this rank's resident frames: rulebook chain (site index, 11 rulebooks) +
    20 sparse convolutions + dense(); C5 adds the backward pass (data + weight gradients) and, at N > 1, the flat
    gradient all-reduce.  Unit of work = frame pairs (2 frames), like every other config."""
    import torch.distributed as dist
    import rslo_amd  # noqa: F401
    from rslo_amd import capi
    import spconv
    capi.lib()
    cfg = args.config
    backward = cfg == "c5"
    frames = 1 if cfg == "c2" else 2 * args.batch
    enc, gen, clouds, max_vox = encoder_setup(cfg, dev, frames, seed0=rank * frames)
    feats, coords = encoder_inputs(gen, clouds, max_vox, dev)
    if dist_on:
        from rslo.utils.distributed_utils import average_gradients, broadcast_params
        broadcast_params(enc, 0)

    # The structure work of the NEXT step -- voxelization of its clouds and the whole rulebook chain, one foreign call
    # (rslo_plan_encoder), no host read -- runs on a side stream while the current step's convolutions run, exactly as in the
    # C3 step (rslo_amd.workload.ExamplePrefetcher; the reference's DataLoader workers play this role on the CPU).  A step
    # therefore does MORE than SURVEY 8d asks of C2 (it also voxelizes); --no-prefetch keeps the resident voxelized inputs and
    # builds the rulebooks inline (Python-issued, with its host reads).
    prefetch = None
    if not args.no_prefetch:
        from rslo_amd import workload

        class _EncoderOnly:      # what ExamplePrefetcher / EncoderPlanner read of a network
            def __init__(self):
                self.middle_feature_extractor, self.voxel_generator, self.training = enc, gen, backward

            def plan_example(self, ex):      # capacity-overflow fallback of the native planner
                B = ex["num_voxels"][0].shape[0]
                cs = [torch.cat([c[:, :1] + t * B, c[:, 1:]], 1) for t, c in enumerate(ex["coordinates"])]
                ex["sparse_plan"] = enc.plan(torch.cat(cs, 0), len(cs) * B, with_pairs=backward)
                return ex
        try:
            prefetch = workload.ExamplePrefetcher(_EncoderOnly(), max_voxels=max_vox, device=dev, depth=2)
            if prefetch.planner is None:
                prefetch.close()
                prefetch = None
        except Exception:
            prefetch = None
    # C2 is an INFERENCE pass (evaluate.py:363-408): eval-mode encoder, one scan per step, replayed from one hipGraph per plan
    # arena (rslo_amd/inference.py) with the structure work of the coming scans on the runner's side stream.
    # RSLO_C2_GRAPH=0 / --no-prefetch: the eager pass of round 4.
    runner = None
    if cfg == "c2" and not args.no_prefetch and os.environ.get("RSLO_C2_GRAPH", "1") != "0":
        from rslo_amd import inference
        enc.eval()

        class _EvalEncoder:
            middle_feature_extractor, voxel_generator, training = enc, gen, False
        runner = inference.EncoderGraphRunner(_EvalEncoder(), max_vox, dev)
        if prefetch is not None:
            prefetch.close()
            prefetch = None
    T_frames = 1 if cfg == "c2" else 2
    dev_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    per_sample = [[dev_clouds[b * T_frames + t] for t in range(T_frames)] for b in range(frames // T_frames)]
    if prefetch is not None:
        for _ in range(prefetch.depth):
            prefetch.submit(per_sample)

    jobs = []
    if runner is not None:
        for _ in range(int(os.environ.get("RSLO_C2_DEPTH", "3"))):
            jobs.append(runner.submit(dev_clouds[0]))
    eager_pass = [False]

    def step():
        if runner is not None:
            job = jobs.pop(0)
            jobs.append(runner.submit(dev_clouds[0]))
            return runner.run(job, graph=not eager_pass[0])[0]
        if prefetch is not None:
            ex = prefetch.get()
            prefetch.submit(per_sample)
            vox, num = ex["_frame_major"]
            plan = ex["sparse_plan"]
            x = capi.vfe_mean(vox, num).requires_grad_(backward)
            bev, cov = enc(x, plan.indices, frames, plan=plan)
        else:
            x = feats.detach().requires_grad_(backward)
            bev, cov = enc(x, coords, frames)
        if backward:
            (bev.square().mean() + cov.square().mean()).backward()
            if dist_on:
                average_gradients(enc, mean=True)
            enc.zero_grad(set_to_none=True)
        return bev

    probe = ConvProbe(capi)
    probe.pmc_workload_ok = False
    use_probe = (not args.no_kernel_events) and rank == 0
    if use_probe:
        probe.install()
    for _ in range(args.warmup):
        step()

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    probe_steps = min(3, args.steps)
    if runner is not None:
        for k_ in runner.stats:
            runner.stats[k_] = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if use_probe and runner is None and i == args.steps - probe_steps:
            probe.enabled = probe.keep_tables = True
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    st_ = dict(runner.stats) if runner is not None else None
    graph_extras = {}
    if runner is not None:
        # the same scans through the same modules issued EAGERLY (what the graph replays): the launch probe's per-kernel numbers
        # come from these passes, and their time is what the graph is compared with
        eager_pass[0] = True
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_eager = max(10, min(50, args.steps))
        for i in range(n_eager):
            if use_probe and i == n_eager - probe_steps:
                probe.enabled = probe.keep_tables = True
            step()
        torch.cuda.synchronize()
        graph_extras = {"runner_host_split": {"helper_thread_ms_per_plan": round(1e3 * st_["helper_s"] / max(st_["plans"], 1), 4),
                                              "ready_wait_ms_per_run": round(1e3 * st_["ready_wait_s"] / max(st_["runs"], 1), 4),
                                              "handshake_wait_included": "issued.wait() is outside"},
                        "ms_per_pass_same_pipeline_eager": round(1e3 * (time.perf_counter() - t1) / n_eager, 4),
                        "inference_path": "eval-mode encoder, one hipGraph per plan arena (%d captured), capacity-laid-out plan "
                                          "with padding rows, structure work of the coming scans on a side stream" % len(runner._graphs)}
        eager_pass[0] = False
    probe.enabled = False
    per_rank = [elapsed]
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    line = None
    if rank == 0:
        roof = None
        if use_probe and probe.records:
            _, roof = probe.summarize(probe_steps)
            probe.uninstall()
        # algorithmic work of the pass: SURVEY.md 8d formula on the measured pair counts (backward = dgrad + wgrad ~ 2x)
        plan = enc.plan(coords, frames)
        convs = [m for seq in (enc.middle_conv, enc.middle_conv_tail, enc.middle_cov_deconv) for m in seq
                 if isinstance(m, spconv.SparseConvolution)]
        byts = fl = 0
        for m in convs:
            rb = plan.indice_dict[m.indice_key]
            tb = rb.nbrT if m.inverse else rb.nbr
            P = int((tb >= 0).sum())
            n_out, K = tb.shape
            byts += P * m.in_channels * 4 + n_out * m.out_channels * 4 + 8 * P + K * m.in_channels * m.out_channels * 4
            fl += 2 * P * m.in_channels * m.out_channels
        mult = 3.0 if backward else 1.0
        ms = 1e3 * elapsed / args.steps
        extras = {}
        if cfg == "c2":       # the same pass on rulebooks planned once, eager and replayed from one hipGraph
            def planned():
                return enc(feats, coords, frames, plan=plan)[0]
            with torch.no_grad():
                for _ in range(5):
                    planned()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    planned()
                torch.cuda.synchronize()
                extras["ms_per_pass_rulebooks_planned_once"] = round(1e3 * (time.perf_counter() - t1) / args.steps, 4)
                try:
                    side = torch.cuda.Stream()
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            planned()
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        out_static = planned()
                    for _ in range(5):
                        graph.replay()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        graph.replay()
                    torch.cuda.synchronize()
                    extras["ms_per_pass_planned_one_hipgraph"] = round(1e3 * (time.perf_counter() - t1) / args.steps, 4)
                    extras["hipgraph_equals_eager"] = bool(torch.equal(out_static, planned()))
                except Exception as e:       # recorded, not fatal
                    extras["hipgraph_error"] = repr(e)[:200]
        pairs_per_step = frames / 2.0
        line = {
            "metric": "frame-pairs/sec %s GU encoder (synthetic %d-ring scans, ~%dk pts)" % (
                "fwd+bwd" if backward else "forward-only", 128 if backward else 64, clouds[0].shape[0] // 1000),
            "value": round(pairs_per_step * world * args.steps / elapsed, 3), "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({
                "workload": ("C2: forward-only GU encoder + covariance branch, 1 x 64-ring scan (%d points -> %d voxels), bs 1, "
                             "fp32; a step = %s + 20 sparse convs + dense() (= half a frame pair)" % (
                                 clouds[0].shape[0], feats.shape[0],
                                 "voxelization + rulebook chain of the coming scans on a side stream (rslo_plan_encoder) + VFE, the pass "
                                 "replayed from one hipGraph per plan arena (eval mode: rslo_amd/inference.py)" if runner is not None else
                                 ("voxelization + rulebook chain of the next frame on a side stream (rslo_plan_encoder) + VFE"
                                  if prefetch is not None else "rulebook chain on a resident voxelized frame"))) if cfg == "c2" else
                            ("C5: GU encoder + covariance branch fwd+bwd, %d x 128-ring scans (%d points/frame, %d voxels in all), "
                             "0.1 m voxels, sparse shape %s, bs %d frame pairs/GPU, fp32, dp%d; the 256-channel BEV map does "
                             "not fit the head (SURVEY 8d): encoder only; %s" % (
                                 frames, clouds[0].shape[0], feats.shape[0], [int(v) for v in enc.sparse_shape], args.batch, world,
                                 "voxelization + rulebooks of the next step on a side stream (rslo_plan_encoder)"
                                 if prefetch is not None else "resident voxelized inputs, rulebooks built inline")),
                "frames_per_step_per_gpu": frames, "voxels": int(feats.shape[0]), "sparse_convs": len(convs),
                "voxelize_and_plan_on_side_stream": prefetch is not None or runner is not None,
                "lib_sha256": lib_hash(),
                "algorithmic_GB_per_step": round(mult * byts / 1e9, 3), "algorithmic_GFLOP_per_step": round(mult * fl / 1e9, 2),
                "whole_pass_algorithmic_GBps": round(mult * byts / ms / 1e6, 1),
                "whole_pass_hbm_roofline_frac": round(mult * byts / ms / 1e6 / HBM_PEAK_GBS, 4),
                "ideal_ms_at_8TBps": round(mult * byts / 8e9, 4)}, **extras, **graph_extras),
            "roofline": roof, "cpu_baseline": None}
        if dist_on:
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = None
            line["rccl"] = {"ranks": world, "version": ver, "backend": dist.get_backend(),
                            "ms_per_step_min": round(1e3 * min(per_rank) / args.steps, 3),
                            "ms_per_step_max": round(1e3 * max(per_rank) / args.steps, 3)}
        if prefetch is not None:
            prefetch.close()
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = encoder_cpu_baseline(args)
            except Exception as e:
                line["cpu_baseline"] = {"error": repr(e)}
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        print(json.dumps(line), flush=True)


def _child_line(r):
    """The JSON line of a child bench.py from its captured stdout (the collectives library writes its banner through C stdio:
    the line may follow it without a newline of its own)."""
    out = r.stdout or ""
    i = out.rfind('{"metric"')
    if i < 0:
        raise RuntimeError("child bench.py printed no result line (rc %s); stderr tail: %s" % (
            r.returncode, (r.stderr or "")[-400:].replace("\n", " | ")))
    return json.loads(out[i:].splitlines()[0])


# --------------------------------------------------------------------------------------------- the step of a rank of an N > 1 job
def multirank_child(args, cores=None, sibling_blocks=None):
    """This script's C3 step in a child process with a ONE-rank RCCL group and the multi-rank code path forced
    (RSLO_FORCE_SYNCBN_PATH=1: SyncBN statistics meet the "peers" through a world-size-1 peer comm, gradient buckets go through
    the overlapped RCCL exchange) -> the child's JSON line.  cores: confine the child to these CPUs (and turn its own pinning
    off); sibling_blocks: CPU lists that are kept busy by interpreter loops for the duration (one process per CPU)."""
    import subprocess
    env = dict(os.environ, RSLO_BENCH_FORCE_DIST="1", RSLO_FORCE_SYNCBN_PATH="1", RSLO_BENCH_MULTIRANK_CHILD="0",
               RSLO_BENCH_OTHER_CONFIGS="0", RSLO_BENCH_STREAM_SPLIT="0", MASTER_PORT=str(_free_port()))
    pre = None
    if cores is not None:
        env["RSLO_BENCH_PIN"] = "0"
        pre = (lambda cs=set(cores): os.sched_setaffinity(0, cs))
    sibs = []
    try:
        for blk in (sibling_blocks or []):
            for c in blk:
                sibs.append(subprocess.Popen(
                    [sys.executable, "-c", "import os\nos.sched_setaffinity(0, {%d})\nx = list(range(4096))\nwhile True:\n    s = sum(x); x = x[1:] + x[:1]" % c],
                    stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "40", "--warmup", "10", "--no-cpu-baseline",
               "--no-kernel-events", "--batch", str(args.batch), "--rings", str(args.rings), "--dtype", args.dtype]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, preexec_fn=pre)
        try:
            return _child_line(r)
        except RuntimeError as first:      # a child that died (seen about once in ten runs, rc -6 from a library thread, before and
            env["MASTER_PORT"] = str(_free_port())      # after round 6's changes; not reproduced stand-alone): run it once more and say so
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, preexec_fn=pre)
            line = _child_line(r)
            line.setdefault("config", {})["child_retried_after"] = str(first)[:300]
            return line
    finally:
        for p_ in sibs:          # exactly the processes started here
            p_.kill()
        for p_ in sibs:
            p_.wait()


# --------------------------------------------------------------------------------------------- the other BASELINE configs
def other_configs(args, affinity=None):
    """BASELINE.json configs[1] (C2), configs[3] (C4, its per-GPU part) and configs[4] (C5, its per-GPU part) from the default
    invocation: one child process of this script per config (`--config ...`, the child's own step counts -- C2 / C5 steps are
    0.7-5 ms, C4's is the C3 step with bf16 operands), each child's line condensed to what the headline line carries for C3.
    `lib_sha256` of every entry is the child's own reading of the source hash of the library it ran."""
    import subprocess
    out = {}
    plan = (("c2", ["--steps", "200", "--warmup", "30"]),
            ("c4", ["--steps", str(max(args.steps, 20)), "--warmup", str(max(args.warmup, 5)), "--no-cpu-baseline"]),
            ("c5", ["--steps", "40", "--warmup", "8"]))
    for cfg, extra in plan:
        t0 = time.time()
        try:
            env = dict(os.environ, RSLO_BENCH_OTHER_CONFIGS="0", RSLO_BENCH_MULTIRANK_CHILD="0", RSLO_BENCH_STREAM_SPLIT="0")
            cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg] + extra
            # (the parent is pinned to 8 cores; a child starts from the host's whole mask -- it pins its own GPU run, and its
            # CPU baseline uses every core, like the headline line's)
            pre = (lambda a=set(affinity): os.sched_setaffinity(0, a)) if affinity else None
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, preexec_fn=pre)
            child = _child_line(r)
            roof = child.get("roofline") or {}
            cpu = child.get("cpu_baseline") or {}
            out[cfg] = {
                "workload": child["config"]["workload"][:160], "metric": child["metric"],
                "value": child["value"], "unit": child["unit"], "ms_per_step": child["ms_per_step"], "dtype": child["dtype"],
                "steps": child["steps"], "warmup": child["warmup"], "lib_sha256": child["config"].get("lib_sha256"),
                "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us",
                                                      "mfma_useful_frac") if k in roof} or None,
                "cpu_baseline": ({k: cpu.get(k) for k in ("value", "unit", "cores", "kind") if k in cpu} or None),
                "whole_pass_hbm_roofline_frac": child["config"].get("whole_pass_hbm_roofline_frac"),
                "ideal_ms_at_8TBps": child["config"].get("ideal_ms_at_8TBps"),
                "child_wall_s": round(time.time() - t0, 1)}
            if cfg == "c4":
                out[cfg]["note"] = ("per-GPU part of configs[3] (8 x MI355X DDP, bs 4 / GPU, bf16 operands with fp32 rulebook "
                                    "indices) at N = 1; cpu_baseline: the C3 line's (the CPU path has no bf16 mode)")
        except Exception as e:       # a failed child is recorded, the headline line is never lost over it
            out[cfg] = {"error": repr(e)[:300], "child_wall_s": round(time.time() - t0, 1)}
    return out


# --------------------------------------------------------------------------------------------- N > 1 launcher
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, one process per GPU
    (the reference does the same from its own entry point: train_hdf5.py:190-256 `multi_proc_train` ->
    `mp.spawn(train_worker, nprocs=gpus_per_node)`, NCCL init at :322-344).  The ranks run this same file under
    torch.distributed.run on 127.0.0.1; rank 0 prints the JSON line, which passes through this process's stdout."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")          # torch.distributed.run would set 1 and say so on stderr
    port = env.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """The launcher, rendezvous and timing protocol of the N-rank run without a GPU: gloo on the CPU, an empty step.
    Checks what a scaling run depends on before any kernel does: every rank started, world == --gpus, the barrier /
    max-over-ranks timing and the per-rank gather work, rank 0 alone prints the line."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    for _ in range(args.warmup):
        time.sleep(1e-3)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": "frame-pairs/sec fwd+bwd (DRY RUN: no kernels ran)", "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "none", "dry_run": True,
            "config": {"workload": "dry run of the launcher / rendezvous / timing protocol (gloo, CPU)", "frame_pairs_per_gpu": args.batch},
            "ranks": {"world": world, "backend": "gloo",
                      "ms_per_step_min": round(1e3 * min(per_rank) / args.steps, 3),
                      "ms_per_step_max": round(1e3 * max(per_rank) / args.steps, 3)},
            "roofline": None, "cpu_baseline": None}), flush=True)


# --------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    if args.cpu_baseline_only:          # child of a pinned run: the CPU baseline on the host's full affinity mask
        sys.path.insert(0, ROOT) if ROOT not in sys.path else None
        import rslo_amd  # noqa: F401
        print(json.dumps(encoder_cpu_baseline(args) if args.config in ("c2", "c5") else cpu_baseline(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # no launcher around us: start the ranks ourselves
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, world, rank)
    dist_on = world > 1 or os.environ.get("RSLO_BENCH_FORCE_DIST", "0") == "1"   # (the latter: DDP smoke test at N=1)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a bench run is minutes long: a peer that never shows up fails the run after 2 minutes, not after the 10-minute
        # production default (rslo_amd/peer.py reads the variable at import, below)
        os.environ.setdefault("RSLO_PEER_TIMEOUT_MS", "120000")
        # RSLO_BENCH_ONE_GPU=1: all ranks share GPU 0 and talk through gloo -- a functional check of the multi-rank
        # code path (SyncBN statistics exchange, gradient all-reduce) on a single-GPU box; not a performance mode
        one_gpu = os.environ.get("RSLO_BENCH_ONE_GPU", "0") == "1"
        if one_gpu:
            local_rank = 0
        elif torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            raise SystemExit("bench.py: %d ranks on this node but only %d GPUs visible (one rank per GPU; RSLO_BENCH_ONE_GPU=1 "
                             "is the functional two-ranks-on-one-GPU mode)" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if world > 1 and not one_gpu:
            # the first real N > 1 run must be diagnosable in one shot: RCCL's INIT / GRAPH lines (rings, trees, transports
            # per channel, xGMI vs PCIe) go to a per-rank file whose relevant lines rank 0 quotes in `rccl.topology_lines`
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
            os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rslo_rccl_%d_rank%%p.log" % os.getppid())
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist_on else 0)

    if args.config in ("c2", "c5"):
        return run_encoder(args, world, rank, local_rank, dist_on, dev)

    pinned = None
    orig_affinity = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if os.environ.get("RSLO_BENCH_PIN", "1") != "0":
        pinned = pin_to_quiet_cores(int(os.environ.get("RSLO_BENCH_PIN_N", "8")),
                                    (local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))))

    import rslo_amd  # noqa: F401
    from rslo_amd import capi, workload
    capi.lib()   # fail loudly if the HIP library is missing

    torch.manual_seed(7)
    net, _ = workload.build_network(device=dev)
    net.train()
    net.global_step.fill_(2000)          # past the warm-up: predicted pose in the loss, icp_iter = 2
    if os.environ.get("RSLO_HEAD_NHWC", "0") == "1":     # experiment: dense head in channels-last
        net.odom_predictor.to(memory_format=torch.channels_last)
        net.odom_predictor.channels_last = True
    model = net
    if dist_on:
        # Data parallel without the DDP wrapper: identical initial weights (broadcast), and after backward ONE flat
        # 48 MB all-reduce of the gradients over RCCL (rslo.utils.distributed_utils.average_gradients).  torch DDP with
        # find_unused_parameters=True (77 of 290 tensors never get a gradient) costs ~8 ms of host time per step here.
        from rslo.utils.distributed_utils import OverlappedGradientExchange, average_gradients, broadcast_params
        broadcast_params(net, 0)
        # the head's ~11 M gradients leave as one asynchronous all-reduce when the head's backward is done, overlapped
        # with the encoder's backward; the rest follows after backward (RSLO_OVERLAP_GRADS=0: one bucket after backward)
        grad_exchange = None
        if os.environ.get("RSLO_OVERLAP_GRADS", "1") != "0":
            grad_exchange = OverlappedGradientExchange(net, net.odom_predictor, mean=True, module_hook=False)
            net.__dict__["_grad_exchange"] = grad_exchange
        # RCCL writes its version banner (NCCL_DEBUG=VERSION on this pool) through C stdio, which would otherwise be
        # flushed at process exit, AFTER the JSON line: push it out now, on every rank
        torch.cuda.synchronize()
        _flush_c_stdio()
    params = [p for p in net.parameters() if p.requires_grad]
    # the reference's training step: 8-group Adam behind OptimWrapper (decoupled weight decay) + OneCycle schedule,
    # built from the shipped train_config (train_hdf5.py:408-411,478-480,618,661-674)
    from rslo.builder import lr_scheduler_builder, optimizer_builder
    from rslo_amd import optim as hip_optim
    from rslo.utils import config_text
    train_cfg = config_text.shipped_config().train_config
    opt = optimizer_builder.build(train_cfg.optimizer, net)
    sched = lr_scheduler_builder.build(train_cfg.optimizer, opt, train_cfg.steps)
    from apex import amp        # the real package if installed, else rslo_amd/compat/apex
    net, opt = amp.initialize(net, opt, opt_level="O1" if args.dtype == "bf16" else "O0")   # train_hdf5.py:456-461

    # RSLO_BENCH_BATCHES distinct resident batches (default 3), used round-robin: consecutive steps voxelize and plan
    # DIFFERENT scans (different voxel counts, tables, pair lists), not one cache-warm batch over and over
    n_sets = max(1, int(os.environ.get("RSLO_BENCH_BATCHES", "3")))
    cloud_sets = []
    for k in range(n_sets):
        cs = workload.kitti_pairs(args.batch, n_el=args.rings, start=(rank * n_sets + k) * args.batch)
        cloud_sets.append([[torch.from_numpy(c).to(dev) for c in pair] for pair in cs])
    clouds = cloud_sets[0]
    submitted = [0]

    def next_clouds():
        submitted[0] += 1
        return cloud_sets[submitted[0] % n_sets]
    fixed_example = workload.make_example(net, clouds, device=dev) if args.no_voxelize else None
    if fixed_example is not None and os.environ.get("RSLO_BENCH_FIXED_PLAN") == "1":
        net.plan_example(fixed_example)      # diagnostic: no structure work at all inside the step (lower bound)

    # the next batch is voxelized and its rulebooks planned on a side stream while the current step runs (what the
    # reference's DataLoader workers do on the CPU); every step still voxelizes its own 2 x batch clouds
    prefetch = None
    if fixed_example is None and not args.no_prefetch:
        # one helper is enough: with 2-3 the wait for the next example goes to zero but backward stretches by the same
        # amount (14.9 / 14.8 / 15.3 ms per step with 1 / 2 / 3 helpers) -- the GPU, shared by both streams, is the bound
        workers = int(os.environ.get("RSLO_PREFETCH_WORKERS", "1"))
        depth = int(os.environ.get("RSLO_PREFETCH_DEPTH", str(max(2, workers + 1))))
        prefetch = workload.ExamplePrefetcher(net, device=dev, depth=depth, workers=workers)
        for _ in range(depth):
            prefetch.submit(next_clouds())

    wait = [0.0]
    # diagnostic for the two-ranks-on-one-GPU functional mode, where the gradient exchange runs over gloo / TCP and hides what
    # the SyncBN exchange costs: ranks then diverge -- never a measurement of the step
    skip_grad_exchange = os.environ.get("RSLO_BENCH_SKIP_GRAD_EXCHANGE", "0") == "1"
    if skip_grad_exchange and dist_on and grad_exchange is not None:
        net.__dict__.pop("_grad_exchange", None)

    phases = os.environ.get("RSLO_BENCH_PHASES") == "1"       # wall / CPU time of the issuing thread per phase (stderr)
    ph = {k: [0.0, 0.0] for k in ("get", "fwd", "bwd", "opt")}

    def mark(name, w0, c0):
        if phases:
            ph[name][0] += time.perf_counter() - w0
            ph[name][1] += time.thread_time() - c0
        return time.perf_counter(), time.thread_time()

    ph_hook = [None]
    gx_events = []       # (before, after) grad_exchange.finish() on the training stream: what of the gradient exchange is NOT hidden

    def step():
        w0, c0 = time.perf_counter(), time.thread_time()
        if ph_hook[0]:
            ph_hook[0]("optimizer_and_step_start")
        if prefetch is not None:
            tw = time.perf_counter()
            ex = prefetch.get()
            wait[0] += time.perf_counter() - tw
        elif fixed_example is not None:
            ex = dict(fixed_example)
        else:
            ex = workload.make_example(net, next_clouds(), device=dev)
        w0, c0 = mark("get", w0, c0)
        sched.step(net.get_global_step())
        opt.zero_grad()
        ret = model(ex)
        if prefetch is not None:
            prefetch.submit(next_clouds())
        w0, c0 = mark("fwd", w0, c0)
        if ph_hook[0]:
            ph_hook[0]("forward")
        with amp.scale_loss(ret["loss"].mean(), opt) as scaled_loss:       # train_hdf5.py:663
            scaled_loss.backward()
        if dist_on and not skip_grad_exchange:
            gx0 = torch.cuda.Event(enable_timing=True)
            gx1 = torch.cuda.Event(enable_timing=True)
            gx0.record()
            if grad_exchange is not None:
                grad_exchange.finish()
            else:
                average_gradients(net, mean=True)
            gx1.record()
            gx_events.append((gx0, gx1))
        w0, c0 = mark("bwd", w0, c0)
        if ph_hook[0]:
            ph_hook[0]("backward")
        if not args.no_optim:
            hip_optim.clip_grad_norm_(params, 10.0, optimizer=opt)      # train_hdf5.py:671 on the optimizer's tables
            opt.step()
            net.update_global_step()
        mark("opt", w0, c0)
        return ret

    # The training stream runs at HIGH priority, the structure work of the coming batches (side stream) at normal
    # priority: its ~250 small integer kernels then take the CU slots the training kernels leave free instead of an
    # equal share of the machine (streams of one priority alternate).  RSLO_TRAIN_PRIORITY=0: both normal (A/B runs).
    if os.environ.get("RSLO_TRAIN_PRIORITY", "1") != "0":
        train_stream = torch.cuda.Stream(dev, priority=min(torch.cuda.Stream.priority_range()))
        train_stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(train_stream)

    probe = ConvProbe(capi)
    use_probe = (not args.no_kernel_events) and rank == 0
    if use_probe:
        probe.install()

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    # per-launch HIP events on the last PROBE_STEPS timed steps only (the events themselves cost host time)
    probe_steps = min(3, args.steps)
    del gx_events[:]
    wait[0] = 0.0
    if prefetch is not None:
        prefetch.plan_wait_seconds = 0.0
    for v in ph.values():
        v[0] = v[1] = 0.0
    alloc0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)      # hipMalloc calls so far (caching allocator misses)
    live0 = torch.cuda.memory_allocated(dev)
    gc_every = int(os.environ.get("RSLO_BENCH_GC", "0"))                  # diagnostic: cyclic GC every n timed steps
    t0 = time.perf_counter()
    from rslo.models import voxel_odom_net as _von0
    lead_wait0, lead_cpu0 = _von0._LEAD_WAIT[0], _von0._LEAD_WAIT[1]
    cpu0 = time.thread_time()           # CPU time of the issuing thread: close to the wall time = host-bound step
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one event per step: spread of the
    marks[0].record()                                                                # step time without extra syncs
    for i in range(args.steps):
        if use_probe and i == args.steps - probe_steps:
            probe.enabled = True
            probe.keep_tables = True
        ret = step()
        marks[i + 1].record()
        if gc_every and (i + 1) % gc_every == 0:
            import gc
            gc.collect()
    cpu_issue = time.thread_time() - cpu0
    from rslo.models import voxel_odom_net as _von
    held_back = _von._LEAD_WAIT[0] - lead_wait0       # wall seconds the issuing thread was held back behind the GPU
    held_back_cpu = _von._LEAD_WAIT[1] - lead_cpu0    # CPU seconds of that wait (a sleeping wait on a blocking-sync event: ~0)
    barrier()
    elapsed = time.perf_counter() - t0
    probe.enabled = False
    per_rank = [elapsed]
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)                      # every rank's own clock around the same K steps
        per_rank = [float(g.item()) for g in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    replicas_identical = None
    if dist_on and world > 1 and not skip_grad_exchange:
        # data parallel invariant: identical weights and BatchNorm running statistics on every rank after K steps of averaged
        # gradients and rank-summed statistics (bit for bit: every rank applies the same update to the same values)
        # (the covariance branch's nn.BatchNorm1d layers are LOCAL BatchNorm in the reference too, rslo/models/middle.py:181-198:
        # their running statistics legitimately differ between ranks and are not part of the invariant)
        import hashlib as _hl
        named = list(net.named_parameters()) + [(n_, b_) for n_, b_ in net.odom_predictor.named_buffers() if b_.is_floating_point()]
        digs = {n_: _hl.sha256(t_.detach().float().cpu().numpy().tobytes()).hexdigest()[:16] for n_, t_ in named}
        gathered = [None] * world
        dist.all_gather_object(gathered, digs)
        differing = [n_ for n_ in digs if len({g_[n_] for g_ in gathered}) != 1]
        replicas_identical = not differing
        if differing and rank == 0:
            print("replicas differ in %d of %d tensors, e.g. %s" % (len(differing), len(digs), differing[:12]), file=sys.stderr)
    syncbn_exchange = None
    if dist_on and (world > 1 or os.environ.get("RSLO_FORCE_SYNCBN_PATH", "0") == "1"):
        from rslo_amd import peer as _peer
        from apex import parallel as _apx
        _comm = _peer.comm_for(None)
        syncbn_exchange = ((_comm.transport + (" (per-channel rendezvous inside the BatchNorm kernel: 1 launch per direction; "
                                                "96x176 maps: statistics -> exchange kernel -> apply)"
                                                if _apx._fused_peer_comm(None) is not None else " (same-stream peer kernel: statistics -> exchange -> apply)"))
                           if _comm is not None else "collective (all_reduce)")
        if _comm is not None:
            _comm.check()          # a peer that missed an exchange poisons the statistics with NaN: fail loudly, never report it
    # N > 1 diagnostics, gathered on every rank (collective calls), reported by rank 0
    gx_exposed = peer_waits = transports = topo = None
    if dist_on:
        torch.cuda.synchronize()
        n_gx = min(len(gx_events), args.steps)
        if n_gx:
            gx_exposed = round(sum(a.elapsed_time(b) for a, b in gx_events[-n_gx:]) / n_gx, 3)
        from rslo_amd import peer as _peer2
        _c2 = _peer2.comm_for(None) if (world > 1 or os.environ.get("RSLO_FORCE_SYNCBN_PATH", "0") == "1") else None
        mine = {"transport": _c2.transport if _c2 is not None else "rccl", "wait": _c2.wait_stats() if _c2 is not None else None}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        transports = [a["transport"] for a in allr]
        peer_waits = [a["wait"] for a in allr]
        dbg = os.environ.get("NCCL_DEBUG_FILE", "")
        if rank == 0 and dbg:
            import glob as _glob
            topo = []
            for f in sorted(_glob.glob(dbg.replace("%p", "*").replace("%h", "*")))[:2]:
                try:
                    with open(f, errors="replace") as fh:
                        keep = [ln.strip()[-200:] for ln in fh if any(k in ln for k in ("Ring", "Tree", "Channel", " via ", "XGMI", "xGMI",
                                                                                         "P2P", "NET/", "nranks", "comm 0x"))]
                    topo += keep[:12]
                except OSError:
                    pass
            topo = topo[:24] or None
    if phases and rank == 0:
        st = torch.cuda.memory_stats(dev)
        print("allocator: %d hipMalloc calls during the %d timed steps, %d MB reserved, %d MB peak allocated, live %d -> %d MB" % (
            st.get("num_device_alloc", 0) - alloc0, args.steps, st.get("reserved_bytes.all.current", 0) >> 20,
            st.get("allocated_bytes.all.peak", 0) >> 20, live0 >> 20, torch.cuda.memory_allocated(dev) >> 20), file=sys.stderr)
        print("phases (wall ms, cpu ms per step): " + ", ".join(
            "%s %.2f/%.2f" % (k, 1e3 * v[0] / args.steps, 1e3 * v[1] / args.steps) for k, v in ph.items()), file=sys.stderr)
    loss_val = float(ret["loss"].detach().mean().item())
    per_step = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    # the launch probe's tables are views of the structure plan's arenas, which later steps recycle: count pairs NOW, before
    # the extra measurement steps below run
    probe_groups = probe_roof = None
    if rank == 0 and use_probe and probe.records:
        probe_groups, probe_roof = probe.summarize(probe_steps)
        probe.uninstall()
        probe.records = []
    # Where the training stream's time goes, from HIP events alone (rslo_amd/streamprobe.py): extra steps AFTER the timed
    # region, every launching library call and every explicit stream join bracketed by events on its stream
    stream_split = None
    if world == 1 and not dist_on and not args.no_kernel_events and os.environ.get("RSLO_BENCH_STREAM_SPLIT", "1") != "0":
        from rslo_amd import streamprobe
        n_split = 3
        tstream = torch.cuda.current_stream(dev)
        step(); torch.cuda.synchronize()
        capi.probe_streams(True)
        streamprobe.mark("begin", tstream)
        for _ in range(n_split):
            step()
        streamprobe.mark("end", tstream)
        stream_split = streamprobe.summarize(tstream, n_split)
        capi.probe_streams(False)
        stream_split["plain_step_ms"] = round(float(np.median(per_step)), 3)
        # how far the issuing threads run ahead of the GPU in the PLAIN step (no probe in the library calls): four points per
        # step, each an event on the training stream + the host clock, both zeroed at a drained stream.  lag = GPU time of
        # the point - host time of the point: ~0 means the GPU executed the phase's last launch as soon as it was issued
        # (the stream had been waiting for the host), a large lag means the host was ahead.
        lagpts = []
        hook = {"pts": None}

        def point(name):
            if hook["pts"] is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(tstream)
                hook["pts"].append((name, time.perf_counter(), ev))
        ph_hook[0] = point
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(tstream)
        tstream.synchronize()
        h0 = time.perf_counter()
        hook["pts"] = lagpts
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ph_hook[0] = None
        agg = {}
        for name, h, ev in lagpts:
            agg.setdefault(name, []).append(ev0.elapsed_time(ev) - 1e3 * (h - h0))
        stream_split["plain_step_gpu_lag_behind_host_ms"] = {k: round(float(np.median(v)), 3) for k, v in agg.items()}
        seq = [(n, 1e3 * (h - h0), ev0.elapsed_time(ev)) for n, h, ev in lagpts]
        dh, dg = {}, {}
        for (n0, h_a, g_a), (n1, h_b, g_b) in zip(seq[:-1], seq[1:]):
            dh.setdefault(n1, []).append(h_b - h_a)
            dg.setdefault(n1, []).append(g_b - g_a)
        # the same lag at every dense / sparse convolution launch of the training stream (one event each, ~150 per step)
        capi.probe_streams(True, points_only=("rslo_conv2d_fwd", "rslo_conv2d_fwd_add", "rslo_spconv_fwd", "rslo_spconv_fwd_split",
                                              "rslo_spconv_dgrad", "rslo_conv2d_fwd_s2", "rslo_conv2d_dgrad_s2_add",
                                              "rslo_bn2d_fwd_local", "rslo_bn2d_bwd_local"))
        streamprobe.mark("begin", tstream)
        for _ in range(5):
            step()
        streamprobe.mark("end", tstream)
        stream_split["plain_step_lag_at_conv_launches"] = streamprobe.lag_profile(tstream)
        capi.probe_streams(False)
        stream_split["plain_step_phase_ms"] = {k: {"host_issue": round(float(np.median(dh[k])), 3),
                                                   "gpu": round(float(np.median(dg[k])), 3)} for k in dh}
    if rank == 0:
        roof = probe_roof
        if probe_groups is not None and args.kernel_report:
            with open(args.kernel_report, "w") as f:
                json.dump({k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                           for k, v in probe_groups.items()}, f, indent=1)
        n_points = int(np.mean([c.shape[0] for cs in cloud_sets for pair in cs for c in pair]))
        line = {
            "metric": "frame-pairs/sec fwd+bwd (synthetic KITTI-shaped ~%dk-pt scans)" % (n_points // 1000),
            "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: full fwd+bwd (voxelize + GU encoder + BEV head/vote + chamfer/ICP loss"
                                   "%s), bs=%d frame pairs/GPU, %s, %d-ring scans (~%d pts/frame), dp%d"
                                   % ("C3" if args.dtype == "f32" else ("C4" if world == 8 else "C4 (per-GPU part)"),
                                      "" if args.no_optim else " + Adam step", args.batch,
                                      "fp32" if args.dtype == "f32" else
                                      "bf16 conv operands / bf16 encoder trunk features, fp32 accumulate + masters (amp O1)",
                                      args.rings, n_points, world),
                       "ms_per_step_gpu_timeline": {"median": round(float(np.median(per_step)), 3),
                                                    "p10": round(float(np.percentile(per_step, 10)), 3),
                                                    "p90": round(float(np.percentile(per_step, 90)), 3)},
                       "lib_sha256": lib_hash(),
                       "frame_pairs_per_gpu": args.batch, "points_per_frame": n_points,
                       "voxelize_in_step": not args.no_voxelize, "voxelize_prefetch_stream": prefetch is not None,
                       "distinct_batches": n_sets, "pinned_cpus": pinned,
                       "optimizer_in_step": not args.no_optim,
                       # CPU time of the issuing thread minus the CPU time it spent waiting behind the GPU (the wait sleeps:
                       # its wall time is `host_held_back_ms_per_step`, its CPU time is close to nothing)
                       "host_issue_ms_per_step": round(1e3 * (cpu_issue - held_back_cpu) / args.steps, 3),
                       # get(): waiting for the helper's result / the plan's event + assembling the example from the arena
                       "prefetch_wait_ms_per_step": round(1e3 * wait[0] / args.steps, 3),
                       # the issuing thread sleeping until the GPU is within RSLO_HOST_LEAD forward passes (slack of a
                       # GPU-bound step; an unbounded lead fills the launch queue and costs ~0.4 ms per step)
                       "host_held_back_ms_per_step": round(1e3 * held_back / args.steps, 3),
                       "prefetch_thread_cpu_ms_per_step": (round(1e3 * prefetch.cpu_seconds / max(prefetch.jobs, 1), 3)
                                                           if prefetch is not None and hasattr(prefetch, "jobs") else None),
                       # the training stream's step split into probed kernel time, explicit stream joins and the rest
                       # (boundaries between dependent launches + unprobed torch kernels); measured on 3 extra steps
                       # with events around every library call, so its window is longer than the plain step
                       "train_stream_split": stream_split,
                       # rslo_amd/headgraph.py: "fwd" = the head's forward replayed from a hipGraph (its launches are then not
                       # among the probed ones: the dense groups of `roofline` are the backward's), None = issued launch by launch
                       "head_graph": _head_graph_mode(net),
                       # random-init weights, W + K optimizer steps: the head still votes a pose metres off, so the two chamfer
                       # searches of the consistency loss scan 2-3 x the tiles they scan on a trained head (0.36 + 0.15 ms per
                       # step here against 0.12-0.19 each, profiles/r06_tail_segment.txt): the timed step is not an easy one
                       "final_loss": round(loss_val, 4),
                       "final_loss_note": "random-init weights after warmup + steps optimizer steps; the pose is still metres "
                                          "off, which makes the loss's two chamfer searches 2-3 x slower than on a trained head"},
            "roofline": roof,
            "cpu_baseline": None,
        }
        if dist_on:
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:          # a build without the binding: say so rather than guess
                ver = None
            line["rccl"] = {"ranks": world, "version": ver, "backend": dist.get_backend(),
                            "ms_per_step_min": round(1e3 * min(per_rank) / args.steps, 3),
                            "ms_per_step_max": round(1e3 * max(per_rank) / args.steps, 3),
                            # how the 90 SyncBN statistics exchanges per step travelled (rslo_amd/peer.py)
                            "syncbn_exchange": syncbn_exchange, "replicas_identical": replicas_identical,
                            # diagnostics for the first real N > 1 run (verdict r4 #6): the part of the gradient exchange
                            # the training stream waits for, per-rank peer transport and per-exchange waits, RCCL's own
                            # topology lines
                            "grad_exchange_exposed_ms_per_step": gx_exposed, "syncbn_wait_us": peer_waits,
                            "syncbn_transport_per_rank": transports, "topology_lines": topo}
        if world == 1 and not dist_on and os.environ.get("RSLO_BENCH_MULTIRANK_CHILD", "1") != "0" and not args.no_kernel_events:
            # The step a rank of an N > 1 job really runs, measured on this one GPU: a child process with a ONE-rank RCCL
            # group and the multi-rank code path forced (RSLO_FORCE_SYNCBN_PATH=1: SyncBN statistics meet the "peers" through
            # a world-size-1 peer comm, gradient buckets go through the overlapped RCCL exchange).  What it cannot show is
            # the peers' arrival skew and the xGMI hop.
            try:
                child = multirank_child(args)
                line["config"]["multirank_path_ms_per_step"] = child["ms_per_step"]
                line["config"]["multirank_path"] = {
                    "ms_per_step": child["ms_per_step"], "plain_ms_per_step": line["ms_per_step"],
                    "ratio_to_plain": round(child["ms_per_step"] / line["ms_per_step"], 4),
                    # round 6: the head's forward is replayed from its hipGraph on this path too (the SyncBN exchanges inside the
                    # capture carry numbers relative to a device word, rslo_amd/headgraph.py)
                    "head_graph": child["config"].get("head_graph"),
                    "host_issue_ms_per_step": child["config"].get("host_issue_ms_per_step"),
                    "syncbn_exchange": (child.get("rccl") or {}).get("syncbn_exchange"),
                    # rslo_amd/streams.py: a process with a process group issues the dense weight gradients on the covariance
                    # branch's stream (the step tolerates four active streams; the fourth is left to the collective library)
                    "side_streams": "shared (covariance branch + dense weight gradients on one stream)",
                    "child_retried_after": child["config"].get("child_retried_after"),
                    "what": "one-rank RCCL group, multi-rank SyncBN path forced (RSLO_FORCE_SYNCBN_PATH=1), overlapped "
                            "gradient exchange on; 40 steps in a child process on the same GPU"}
                # what the HOST costs a rank of an 8-rank job: the same child confined to 2 cores, alone and beside 7 sibling
                # processes that keep the neighbouring 2-core blocks busy with interpreter work (CPU-only contention: caches,
                # memory, the socket's power budget; the GPU is this child's alone)
                if pinned is not None and os.environ.get("RSLO_BENCH_HOST_PRICE", "1") != "0":
                    allowed = sorted(orig_affinity)
                    blk = [c for c in range(pinned[0], pinned[0] + 16) if c in allowed]
                    if len(blk) == 16:
                        try:
                            two = multirank_child(args, cores=blk[:2])
                            sib = multirank_child(args, cores=blk[:2], sibling_blocks=[blk[2 * k:2 * k + 2] for k in range(1, 8)])
                            line["config"]["multirank_path"]["two_cores_per_rank"] = {
                                "ms_per_step": two["ms_per_step"],
                                "host_issue_ms_per_step": two["config"].get("host_issue_ms_per_step"),
                                "with_7_busy_siblings": {"ms_per_step": sib["ms_per_step"],
                                                         "host_issue_ms_per_step": sib["config"].get("host_issue_ms_per_step")},
                                "cores": blk[:2]}
                        except Exception as e:
                            line["config"]["multirank_path"]["two_cores_per_rank"] = {"error": repr(e)[:400]}
            except Exception as e:
                line["config"]["multirank_path"] = dict(line["config"].get("multirank_path") or {}, error=repr(e))
        if (world == 1 and not dist_on and args.config == "c3" and args.batch == 4 and args.rings == 64
                and not args.no_cpu_baseline and not args.no_kernel_events      # (the plain default invocation only)
                and os.environ.get("RSLO_BENCH_OTHER_CONFIGS", "1") != "0"):
            # BASELINE.json configs[1], [3] (per-GPU part) and [4] measured by the SAME invocation the driver runs, each in a
            # child process of this script (`--config c2 | c4 | c5`: its own timing protocol, roofline and CPU baseline), and
            # carried under config.other_configs of the one final line.  The C3 headline above is already measured.
            line["config"]["other_configs"] = other_configs(args, orig_affinity)
        if world == 1 and not args.no_cpu_baseline:
            try:
                if pinned is not None:
                    # the GPU run is pinned to 8 cores; the CPU baseline gets the whole host: a child process with the
                    # original affinity mask (OpenMP pools created under the pinned mask would stay on those cores)
                    import subprocess
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--rings",
                                        str(args.rings)], capture_output=True, text=True, timeout=600,
                                       preexec_fn=lambda: os.sched_setaffinity(0, orig_affinity))
                    line["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
                else:
                    line["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:   # the baseline is informational; never lose the measurement over it
                line["cpu_baseline"] = {"error": repr(e)}
    if dist_on:
        dist.barrier()
        from rslo_amd import peer as _peer
        _peer.shutdown()             # unmap the peers' slices / unlink the shared segment before the group goes away
        dist.destroy_process_group()
    _flush_c_stdio()                 # anything the collectives library still holds goes out before the result
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
