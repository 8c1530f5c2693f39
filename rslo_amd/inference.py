"""Frame-after-frame inference of the sparse GU encoder from ONE hipGraph per plan arena.

evaluate.py:363-408 runs the same eval-mode forward for every frame of a sequence (rslo/models/middle.py:219-245 for the
encoder).  Issued eagerly that pass is ~40 dependent launches on levels of 2.6 k .. 31 k rows: 0.6 ms of launch chain for
54 us of roofline work, plus the interpreter time of building a plan's tensor views for every frame.  Here

  * the structure work of a frame (voxelization + every rulebook: rslo_plan_encoder, one foreign call) goes into a
    CAPACITY-laid-out arena whose rows past each level's count are padding rows (rslo_plan_encoder_pad_tails), so every
    pointer and every shape the encoder's kernels see depends on the arena only, not on the scan;
  * VFE + all convolutions + dense() of an arena are captured once into a hipGraph (torch.cuda.graph over the same modules,
    the same kernels: capi launches go to the capture stream) and REPLAYED for every later frame planned into that arena:
    one launch of the graph per frame, counts stay on the device, no host read anywhere;
  * the coming frames' structure work is issued by a helper thread on a side stream while the current frame's graph runs.

Outputs of the VALID rows (and the whole BEV map: padding rows are never scattered) equal the eager pass bit for bit:
same kernels, and the tilings that depend on a level's size are pinned to what the eager pass of a single scan picks
(`rslo_spconv_set_tiling(1, 4)`: 16-row tiles shared by four waves, csrc/spconv.hip).
"""
import queue
import threading

import torch

from rslo_amd import capi
from rslo_amd.plan import EncoderPlanner


class _Handle:
    __slots__ = ("slot", "job", "error", "issued")

    def __init__(self, slot):
        self.slot, self.job, self.error, self.issued = slot, None, None, threading.Event()


class EncoderGraphRunner:
    """runner = EncoderGraphRunner(net_like)       # .middle_feature_extractor (eval mode), .voxel_generator
       job = runner.submit(cloud)                   # structure work of one scan, on the runner's side stream
       bev, cov, n_rows = runner.run(job)           # graph replay on the current stream; cov[:n_rows] are the valid rows
    `bev` / `cov` are static tensors of the job's arena: consume (or copy) them before that arena's next run()."""

    def __init__(self, net_like, max_voxels, device="cuda", point_capacity=160000, arenas=4, frames_per_job=1):
        self.net = net_like
        self.enc = net_like.middle_feature_extractor
        if self.enc.training:
            raise capi.RsloHipError("EncoderGraphRunner is an inference path: put the encoder in eval() mode")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.planner = EncoderPlanner(net_like, max_voxels, arenas=arenas)
        self.point_capacity = int(point_capacity)
        self.frames = int(frames_per_job)
        # structure stream(s), used round-robin.  One is the measured choice: a plan alone takes 0.43 ms per scan on one stream and
        # 0.26 on two alternating ones, but beside the replayed pass two or three streams change nothing (0.75-0.77 ms per scan)
        import os
        prio = os.environ.get("RSLO_INFER_PLAN_PRIORITY", "")
        self.sides = [torch.cuda.Stream(self.device, **({"priority": int(prio)} if prio else {}))
                      for _ in range(int(os.environ.get("RSLO_INFER_PLAN_STREAMS", "1")))]
        self._exact_planner = EncoderPlanner(net_like, max_voxels, arenas=2)
        self.fallbacks = 0
        self._seq = 0
        self._queue = queue.Queue()
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()
        self.stats = {"plans": 0, "helper_s": 0.0, "ready_wait_s": 0.0, "runs": 0}      # host-side time split (bench.py c2)
        self._graphs = {}          # arena slot -> (graph, bev, cov, rows_dev0, arena data_ptr)
        self._last_use = {}        # arena slot -> event recorded behind the last replay that read the arena
        self._pending = {}         # arena slot -> scan submitted into it and not yet run()

    def submit(self, clouds):
        """clouds: one CUDA fp32 [P,F] tensor (or a list of `frames_per_job` of them).  Returns a handle for run().  The
        structure work is issued by the runner's helper thread (one foreign call of ~60 launches: 0.3-0.5 ms of host time
        that would otherwise sit in front of every replay on the calling thread)."""
        if torch.is_tensor(clouds):
            clouds = [clouds]
        slot = self._seq
        a = slot % self.planner.n_arenas
        if a in self._pending:
            # the plan of this scan would overwrite tables the pending replay of handle `self._pending[a]` still has to read
            raise capi.RsloHipError("EncoderGraphRunner.submit: arena %d still holds scan %d, which has not been run(); keep fewer "
                                    "than %d handles outstanding" % (a, self._pending[a], self.planner.n_arenas))
        self._seq += 1
        self._pending[a] = slot
        h = _Handle(slot)
        prev = self._last_use.get(a)
        self._queue.put((h, clouds, prev))
        return h

    def _work(self):
        torch.cuda.set_device(self.device)
        while True:
            item = self._queue.get()
            if item is None:
                return
            h, clouds, prev = item
            import time as _t
            t0 = _t.perf_counter()
            try:
                side = self.sides[h.slot % len(self.sides)]
                with torch.cuda.stream(side):
                    if prev is not None:
                        side.wait_event(prev)               # the arena's previous frame has been consumed on the GPU
                    h.job = self.planner.submit([[c] for c in clouds], with_pairs=False, slot=h.slot,
                                                point_capacity=self.point_capacity)
            except Exception as e:      # surfaces in run()
                h.error = e
            self.stats["plans"] += 1
            self.stats["helper_s"] += _t.perf_counter() - t0
            h.issued.set()

    def close(self):
        self._queue.put(None)
        self._thread.join(timeout=10)

    def _forward(self, job):
        vox, num, plan, rows_dev = self.planner.finish_static(job)
        x = capi.vfe_mean(vox, num)
        bev, cov = self.enc(x, plan.indices, job.n_clouds, plan=plan)
        return bev, cov, rows_dev[0]

    def _exact(self, job):
        """Exact-size plan + eager pass of a scan whose static plan overflowed a level's capacity (fresh tensors, no graph)."""
        ex = self._exact_planner.finish(self._exact_planner.submit(job.clouds, with_pairs=False))
        vox, num = ex["_frame_major"]
        with torch.no_grad():
            bev, cov = self.enc(capi.vfe_mean(vox, num), ex["sparse_plan"].indices, job.n_clouds, plan=ex["sparse_plan"])
        return bev, cov, torch.tensor([cov.shape[0]], dtype=torch.int32, device=cov.device)

    def _capture(self, job, cur):
        """Warm up on a side stream (lazy initialisation inside the modules must not be captured), then capture the pass over
        the job's arena.  The tilings that depend on a level's size are pinned to the single-scan choice (module docstring)."""
        with torch.no_grad(), capi.tuning(spconv_rbw=1, spconv_ks=4):
            s = torch.cuda.Stream(self.device)
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(2):
                    self._forward(job)
            cur.wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                bev, cov, n0 = self._forward(job)
        return g, bev, cov, n0, job.arena.data_ptr()

    def run(self, handle, graph=True):
        """The encoder pass of a submitted scan on the current stream: a replay of its arena's graph (captured on first use),
        or, graph=False, the same modules issued eagerly over the same capacity-laid-out plan."""
        if not handle.issued.wait(timeout=60.0):
            raise capi.RsloHipError("EncoderGraphRunner: the helper thread did not issue the scan's structure work")
        if handle.error is not None:
            raise handle.error
        job = handle.job
        if self._pending.get(handle.slot % self.planner.n_arenas) == handle.slot:
            del self._pending[handle.slot % self.planner.n_arenas]
        cur = torch.cuda.current_stream(self.device)
        # the counts block of the job reached pinned memory behind job.ready; it was submitted `depth` scans ago, so the event has
        # normally passed and this is a read of host memory, not a wait.  A level that outgrew its capacity (not a LiDAR-shaped
        # scan) cannot be replayed: that scan takes the exact-size eager pass.
        import time as _t
        t0 = _t.perf_counter()
        job.ready.synchronize()
        self.stats["ready_wait_s"] += _t.perf_counter() - t0
        self.stats["runs"] += 1
        if int(job.counts[capi.PLAN_CNT_OVERFLOW]) != 0:
            self.fallbacks += 1
            return self._exact(job)
        a = job.slot % self.planner.n_arenas
        if graph:
            ent = self._graphs.get(a)
            if ent is None or ent[4] != job.arena.data_ptr():
                ent = self._graphs[a] = self._capture(job, cur)
            ent[0].replay()
            out = ent[1], ent[2], ent[3]
        else:
            with torch.no_grad(), capi.tuning(spconv_rbw=1, spconv_ks=4):
                out = self._forward(job)
        ev = torch.cuda.Event()
        ev.record(cur)
        self._last_use[a] = ev
        return out
