"""Synthetic workloads for bench.py / smoke() / the parity tests: KITTI-shaped frame pairs turned into the
`example` dict the reference's collate + example_convert_to_torch produce (train_hdf5.py:44-89,
rslo/data/preprocess.py:46-134), with the voxelization done by the network's own voxel generator."""
import numpy as np
import torch

from rslo_amd import synthetic


def build_network(testing=False, device="cuda"):
    import rslo_amd  # noqa: F401
    from rslo.builder import second_builder, voxel_builder
    from rslo.utils import config_text
    cfg = config_text.shipped_config()
    m = cfg.model.second
    vg = voxel_builder.build(m.voxel_generator)
    net = second_builder.build(m, vg, testing=testing)
    return net.to(device), cfg


def make_example(net, clouds_per_sample, max_voxels=synthetic.MAX_VOXELS, device="cuda"):
    """clouds_per_sample: list (batch) of lists (frames) of [P,7] float32 arrays.
    Returns the example dict: per frame t, the B samples' voxels concatenated with the batch index
    prepended to the coordinates (merge_second_batch, preprocess.py:75-89)."""
    B, T = len(clouds_per_sample), len(clouds_per_sample[0])
    ex = {"voxels": [], "num_points": [], "coordinates": [], "num_voxels": []}
    flat = []
    for t in range(T):
        for b in range(B):
            pts = clouds_per_sample[b][t]
            if isinstance(pts, np.ndarray):
                pts = torch.from_numpy(pts)
            flat.append(pts.to(device))
    if hasattr(net.voxel_generator, "generate_many") and flat[0].is_cuda:
        results = net.voxel_generator.generate_many(flat, max_voxels)      # one host read for all clouds
    else:
        results = []
        for pts in flat:
            r = net.voxel_generator.generate(pts, max_voxels)
            results.append((r["voxels"], r["coordinates"], r["num_points_per_voxel"]))
    for t in range(T):
        vs, ns, cs, nv = [], [], [], []
        for b in range(B):
            v, c, n = results[t * B + b]
            vs.append(v)
            ns.append(n)
            cs.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32, device=c.device), c], 1))
            nv.append(v.shape[0])
        ex["voxels"].append(torch.cat(vs, 0))
        ex["num_points"].append(torch.cat(ns, 0))
        ex["coordinates"].append(torch.cat(cs, 0))
        ex["num_voxels"].append(torch.tensor(nv, dtype=torch.int64).reshape(B, 1))
    npairs = T * (T - 1) // 2
    ex["icp_odometry"] = torch.zeros(B * npairs, 7, device=device)
    ex["tq_maps"] = [torch.zeros(B * npairs, 7, 96, 176, device=device)]
    return ex


def kitti_pairs(batch, n_el=64, start=0):
    """`batch` synthetic frame pairs (2 frames each) -> clouds_per_sample."""
    out = []
    for b in range(batch):
        p0, p1, _ = synthetic.frame_pair(start + b, n_el=n_el)
        out.append([p0, p1])
    return out


class ExamplePrefetcher:
    """Voxelization + rulebook planning of the NEXT batch on a side stream / helper thread while the current step runs
    (the role the reference gives its DataLoader workers, which voxelize on the CPU: rslo/data/preprocess.py:461-512;
    here the voxelizer is a GPU kernel, so "loading ahead" means a second HIP stream).  All host reads of data-dependent
    sizes (voxel counts, strided-conv output counts) then wait on the side stream only; the training stream never
    drains.

        pf = ExamplePrefetcher(net); pf.submit(clouds)
        loop:  ex = pf.get();  out = net(ex);  pf.submit(next_clouds);  out["loss"].backward(); ...

    submit() right before backward(): the helper needs the Python interpreter lock only while the main thread sits in
    the C++ autograd engine.  The depth + 1 most recent examples are kept alive so memory handed to the training stream
    is not recycled by the side stream while still in use."""

    def __init__(self, net, max_voxels=synthetic.MAX_VOXELS, device="cuda", plan=True, depth=1, workers=1):
        import os
        import queue
        import threading
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.net, self.max_voxels, self.device, self.plan = net, max_voxels, device, plan
        # native planner (rslo_plan_encoder): voxelization + all rulebooks in one foreign call without a host read; the
        # example is assembled from the arena in get().  RSLO_NATIVE_PLAN=0 keeps the Python-issued form (A/B runs).
        self.planner = None
        if plan and os.environ.get("RSLO_NATIVE_PLAN", "1") != "0":
            from rslo_amd import plan as native_plan
            try:
                self.planner = native_plan.EncoderPlanner(net, max_voxels, arenas=max(4, int(depth) + 3))
            except Exception:      # an encoder that is not a chain of levels: the Python planner handles it
                self.planner = None
        # depth = examples prepared ahead of the one in use: submit() `depth` times before the first get().  With 2 the
        # example of step i+1 is already finished when step i ends, so a late helper thread no longer stalls the step.
        # workers = helper threads, each with its own side stream: one job is a chain of ~7 host reads of data-dependent
        # sizes, each waiting for small kernels that share the GPU with the training stream, so its LATENCY (10-14 ms)
        # can exceed a training step even though its work is 2-3 ms; two jobs in flight hide that (results are handed
        # out in submission order).
        self.depth = max(1, int(depth))
        self.workers = max(1, int(workers))
        self._in = queue.Queue()
        self._cv = threading.Condition()
        self._results = {}          # seq -> (example, ready event, error)
        self._keep = {}             # seq -> example, until the training stream is done with it
        self._next_submit = self._next_get = 0
        self._done_ring = []
        # where on the training stream's timeline a plan job may START (it then runs at low priority beside whatever follows):
        # "none" = at once (round 3), "loss" (default, round 5: 11.50-11.60 vs 11.77-11.79 ms per step) = where the current
        # step's loss begins, "head" = where the head's small-map stages begin, "fwd_end" = behind the current step's loss
        self._gate = os.environ.get("RSLO_PLAN_GATE", "loss")
        self.lead_wait_seconds = self.plan_wait_seconds = 0.0
        self.cpu_seconds, self.jobs = 0.0, 0
        # Several Python threads issue GPU work here.  With the interpreter's default 5 ms switch interval a helper can
        # hold the lock for a third of a step while the training thread's queue runs dry; 0.2 ms keeps all streams
        # fed (measured: 238 -> 257 frame-pairs/s on the same box, back-to-back runs).
        import os
        import sys
        sys.setswitchinterval(float(os.environ.get("RSLO_SWITCH_INTERVAL", "0.0002")))
        pr = os.environ.get("RSLO_PREFETCH_PRIORITY")
        if pr is None:
            try:
                pr = max(torch.cuda.Stream.priority_range())      # (lowest, highest): larger number = lower priority
            except Exception:
                pr = 0
        self.streams = [torch.cuda.Stream(self.device, priority=int(pr)) for _ in range(self.workers)]
        self.stream = self.streams[0]
        self._threads = [threading.Thread(target=self._work, args=(w,), daemon=True) for w in range(self.workers)]
        for t in self._threads:
            t.start()

    def _work(self, w):
        import time
        torch.cuda.set_device(self.device)
        stream = self.streams[w]
        while True:
            job = self._in.get()
            if job is None:
                return
            seq, clouds, prev_done = job
            c0 = time.thread_time()
            try:
                if self.planner is not None:
                    with torch.cuda.stream(stream):
                        for ev_ in (prev_done if isinstance(prev_done, tuple) else (prev_done,)):
                            if ev_ is not None:       # arena reuse (+ the start gate): ordered behind the training stream on the GPU
                                stream.wait_event(ev_)
                        cl = [[c if torch.is_tensor(c) else torch.from_numpy(c) for c in s] for s in clouds]
                        cl = [[c.to(self.device, torch.float32).contiguous() for c in s] for s in cl]
                        ex = self.planner.submit(cl, slot=seq)          # a job: becomes the example in get()
                    ready = ex.ready
                else:
                    if prev_done is not None:
                        prev_done.synchronize()  # the training stream has issued (and finished) everything up to the
                    with self._cv:               # forward of step seq - depth: older examples are no longer read
                        for old in [k for k in self._keep if k <= seq - self.depth - 1]:
                            del self._keep[old]
                    with torch.cuda.stream(stream):
                        ex = make_example(self.net, clouds, self.max_voxels, self.device)
                        if self.plan:
                            self.net.plan_example(ex)
                        ready = torch.cuda.Event()
                        ready.record(stream)
                res = (ex, ready, None)
            except Exception as e:      # surface in get()
                ex, res = None, (None, None, e)
            with self._cv:
                if ex is not None and self.planner is None:
                    self._keep[seq] = ex
                self._results[seq] = res
                self.cpu_seconds += time.thread_time() - c0
                self.jobs += 1
                self._cv.notify_all()

    def _gate_event(self, which):
        try:
            if which == "loss":
                from rslo.models.voxel_odom_net import GATE_EVENTS
                return GATE_EVENTS.get(self.net)
            from rslo.models.odom_pred import SMALL_MAPS_GATE
            return SMALL_MAPS_GATE.get(getattr(self.net, "odom_predictor", None))
        except (ImportError, TypeError):
            return None

    def submit(self, clouds):
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        if self.planner is not None:
            # Arena reuse is the only thing a native-plan job has to wait for.  Job j writes arena j mod A, last read by
            # the step that consumed job j - A; the event of submit j - (A - 1 - depth) was recorded behind the forward
            # of the step AFTER that one, so it covers its backward -- and it is `A - 1 - depth` (= 2) steps old: the
            # GPU has passed it, the job starts at once and has depth steps of GPU time before anyone needs it.
            # (Waiting on the event of THIS submit holds every job back until the current forward has drained on the GPU.)
            self._done_ring.append(done)
            lag = self.planner.n_arenas - 1 - self.depth
            gate = self._gate
            # the arena-reuse event is ALWAYS waited for (it is old: the wait costs nothing); a gate event only moves the start
            # later -- it may be stale (recorded in an earlier step: the head's gate is not re-recorded while the head is
            # replayed from its hipGraph, the loss gate not in eval loops), and a stale event alone orders nothing
            ring_ev = self._done_ring[-1 - lag] if len(self._done_ring) > lag else None
            if gate == "fwd_end":         # start when the forward + loss of the step that just issued them have drained
                done = (ring_ev, done)
            elif gate == "loss" and self._gate_event("loss") is not None:
                done = (ring_ev, self._gate_event("loss"))          # start where this step's loss begins on the GPU
            elif gate == "head" and self._gate_event("head") is not None:
                done = (ring_ev, self._gate_event("head"))          # the head's small-map stages (forward)
            else:
                done = (ring_ev,)
            del self._done_ring[:-(lag + 1)]
        self._in.put((self._next_submit, clouds, done))
        self._next_submit += 1

    def get(self):
        seq = self._next_get
        self._next_get += 1
        with self._cv:
            while seq not in self._results:
                if not self._cv.wait(timeout=5.0) and not any(t.is_alive() for t in self._threads):
                    raise RuntimeError("ExamplePrefetcher: the helper threads died")
            ex, ready, err = self._results.pop(seq)
        if err is not None:
            raise err
        import time
        t0 = time.perf_counter()
        t1 = t0         # (the bound on the issuing thread's lead lives in the network's forward: voxel_odom_net.py)
        if self.planner is not None:
            ex = self.planner.finish(ex)        # host wait on an event recorded a step ago, then arena views
        self.plan_wait_seconds += time.perf_counter() - t1   # the plan itself was late / assembling the example
        from rslo_amd import streamprobe
        cur = torch.cuda.current_stream(self.device)
        streamprobe.wait("structure_plan_of_this_batch", cur, lambda: cur.wait_event(ready))
        return ex

    def close(self):
        for _ in self._threads:
            self._in.put(None)
        for t in self._threads:
            t.join(timeout=10)
