"""Optimizer step of the training path on the hand-written kernels of csrc/optim.hip.

The reference driver clips the global gradient norm and steps a fastai OptimWrapper around torch.optim.Adam
(train_hdf5.py:671-672; rslo/torchplus/train/fastai_optim.py:176-187).  Through torch that is ~25 multi-tensor launches
and ~1.5 ms of interpreter time per step; `AdamStepper` drives the same arithmetic over all parameter tensors with a
tensor table + chunk table on the device: 2 launches for the clip, 1 for decay + Adam.  The state stays in
torch.optim.Adam's own dict (exp_avg, exp_avg_sq, step), so state_dict / checkpoints are unchanged.

Used by torchplus.train.fastai_optim.OptimWrapper.step() when every parameter is a contiguous fp32 CUDA tensor and the
inner optimizer is a plain Adam (RSLO_HIP_OPTIM=0 turns it off); `clip_grad_norm_` below is the drop-in for
torch.nn.utils.clip_grad_norm_ on the same tables.  There is no CPU path here: on CPU tensors the callers keep torch's.
"""
import ctypes as C
import os

import numpy as np
import torch

ENABLED = os.environ.get("RSLO_HIP_OPTIM", "1") != "0"
CHUNK = 4096            # elements per workgroup (multiple of 4: 16-byte accesses)
MAX_GROUPS = 16

_TENSOR_DT = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("step", "<u8"),
                       ("group", "<i4"), ("step_offset", "<i4")])
_CHUNK_DT = np.dtype([("tensor", "<i4"), ("count", "<i4"), ("offset", "<i8")])


class _Group(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double)]


class _Hyper(C.Structure):
    _fields_ = [("group", _Group * MAX_GROUPS)]


def _usable(p):
    return p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() % 16 == 0


class AdamStepper:
    """Tables for one torch.optim.Adam instance over all trainable tensors, param_groups order.  A tensor without a
    gradient in a step (77 of the network's 290 never get one: SURVEY App-A.2) is left out of the norm and of the moment
    update but still decays, as under the reference's wrapper."""

    def __init__(self, opt):
        self.opt = opt
        self.key = None           # ids of the tensors of the current tables
        self.step_count = None
        self._ring = []           # pinned host tables in rotation (an async upload may still be reading the previous one)
        self._ring_pos = 0
        # load_state_dict replaces the moment tensors: rebuild the tables afterwards
        if hasattr(opt, "register_load_state_dict_post_hook"):
            opt.register_load_state_dict_post_hook(lambda o: setattr(self, "key", None))

    # ------------------------------------------------------------------ eligibility
    @staticmethod
    def supports(opt):
        if not ENABLED or type(opt) is not torch.optim.Adam or len(opt.param_groups) > MAX_GROUPS:
            return False
        for g in opt.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
                return False
            if g.get("decoupled_weight_decay"):
                return False
            if isinstance(g["lr"], torch.Tensor) or not all(_usable(p) for p in g["params"]):
                return False
        return any(len(g["params"]) for g in opt.param_groups)

    # ------------------------------------------------------------------ tables
    def _all(self):
        return [(gi, p) for gi, g in enumerate(self.opt.param_groups) for p in g["params"] if p.requires_grad]

    def _init_state(self, p):
        """What torch.optim.Adam._init_group creates for a fused optimizer."""
        st = self.opt.state[p]
        st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _build(self, tensors):
        """Tables over ALL trainable tensors (the decoupled decay touches every one of them; the moment update only those
        that have a gradient in the step at hand)."""
        dev = tensors[0][1].device
        state = self.opt.state      # a defaultdict: read it with .get(), or never-stepped parameters gain empty entries
        with_state = []
        for gi, p in tensors:
            st = state.get(p)
            if not st:
                if p.grad is None:
                    continue
                self._init_state(p)
                st = state[p]
            if not (_usable(st["exp_avg"]) and _usable(st["exp_avg_sq"])):
                return False
            if not (isinstance(st["step"], torch.Tensor) and st["step"].is_cuda and st["step"].dtype == torch.float32):
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p.device)
            with_state.append(p)
        steps = {}
        if with_state:      # one host read, at (re)build time only
            steps = dict(zip((id(p) for p in with_state),
                             (int(v) for v in torch.stack([state[p]["step"] for p in with_state]).tolist())))
        # torch counts steps per tensor; the kernel takes ONE count plus a per-tensor offset (tensors that joined later
        # or skipped steps lag behind)
        self.step_count = max(steps.values()) if steps else 0
        host = np.zeros(len(tensors), _TENSOR_DT)
        chunks = []
        for i, (gi, p) in enumerate(tensors):
            st = state.get(p) or {}
            has = len(st) > 0
            host[i] = (p.data_ptr(), 0, st["exp_avg"].data_ptr() if has else 0, st["exp_avg_sq"].data_ptr() if has else 0,
                       st["step"].data_ptr() if has else 0, gi, steps[id(p)] - self.step_count if has else 0)
            numel = p.numel()
            for off in range(0, numel, CHUNK):
                chunks.append((i, min(CHUNK, numel - off), off))
        ch = np.array(chunks, _CHUNK_DT)
        self.host = host
        self.n_chunks = len(ch)
        self.tensors_dev = torch.empty(host.nbytes, dtype=torch.uint8, device=dev)
        self.chunks_dev = torch.from_numpy(ch.view(np.uint8).copy()).to(dev)
        self.partial = torch.empty(self.n_chunks, dtype=torch.float64, device=dev)
        self.total_norm = torch.zeros((), dtype=torch.float32, device=dev)
        self._ring = [torch.empty(host.nbytes, dtype=torch.uint8).pin_memory() for _ in range(4)]
        self.grad_ptrs = None
        self.tensors = tensors
        self.has_state = np.array([bool(state.get(p)) for _, p in tensors])
        self.key = tuple(id(p) for _, p in tensors)
        return True

    def _refresh(self):
        """Tables with this step's gradient pointers; False if this step has to go through torch."""
        tensors = self._all()
        if not tensors:
            return False
        if self.key is not None:      # cheap staleness probe (someone replaced opt.state behind our back)
            probe = int(np.argmax(self.has_state)) if self.has_state.any() else None
            if probe is not None:
                st0 = self.opt.state.get(self.tensors[probe][1], {})
                if "exp_avg" not in st0 or st0["exp_avg"].data_ptr() != int(self.host["exp_avg"][probe]):
                    self.key = None
        if self.key != tuple(id(p) for _, p in tensors):
            if not self._build(tensors):
                self.key = None
                return False
        ptrs = np.zeros(len(tensors), dtype=np.uint64)
        for i, (_, p) in enumerate(tensors):
            g = p.grad
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous() or not g.is_cuda:
                return self._torch_step("a gradient that is not a contiguous fp32 CUDA tensor")
            ptrs[i] = g.data_ptr()
        if (ptrs % 16).any():
            return self._torch_step("a gradient tensor that is not 16-byte aligned")
        late = (ptrs != 0) & ~self.has_state
        if late.any():      # first gradient of a tensor: torch starts ITS count at 1 whatever the others have reached
            for i in np.nonzero(late)[0]:
                p = self.tensors[i][1]
                self._init_state(p)
                st = self.opt.state[p]
                self.host["exp_avg"][i], self.host["exp_avg_sq"][i] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                self.host["step"][i] = st["step"].data_ptr()
                self.host["step_offset"][i] = -self.step_count
                self.has_state[i] = True
            self.grad_ptrs = None
        if self.grad_ptrs is None or not np.array_equal(ptrs, self.grad_ptrs):
            self.host["grad"] = ptrs
            pin = self._ring[self._ring_pos]
            self._ring_pos = (self._ring_pos + 1) % len(self._ring)
            pin.numpy()[:] = self.host.view(np.uint8)
            self.tensors_dev.copy_(pin, non_blocking=True)
            self.grad_ptrs = ptrs
        return bool(ptrs.any())

    def _torch_step(self, why):
        """This step goes through torch.optim (same update, ~30 launches instead of 2): said once, because it is slow."""
        if not self.__dict__.get("_warned"):
            import warnings
            warnings.warn("rslo_amd.optim: the fused clip + Adam kernels are skipped for this step (%s); torch's own "
                          "optimizer path runs instead" % why)
            self._warned = True
        return False

    # ------------------------------------------------------------------ the two operations
    def clip_grad_norm_(self, max_norm):
        """-> total norm (device scalar, no host read) or None when the tables cannot serve this step."""
        from rslo_amd import capi
        if not self._refresh():
            return None
        capi._chk(capi.lib().rslo_opt_clip_grad_norm(self.tensors_dev.data_ptr(), self.chunks_dev.data_ptr(), self.n_chunks,
                                                     float(max_norm), self.partial.data_ptr(), self.total_norm.data_ptr(),
                                                     capi._stream()), "rslo_opt_clip_grad_norm")
        return self.total_norm

    def step(self, decoupled_wd=None):
        """One Adam update of every tensor that has a gradient; decoupled_wd: per param group (None = 0).
        Returns False (nothing done) when this step has to go through torch."""
        from rslo_amd import capi
        if not self._refresh():
            self.key = None             # torch steps instead and advances the device-side counts: re-read them next time
            return False
        hyper = _Hyper()
        for gi, g in enumerate(self.opt.param_groups):
            if g["weight_decay"] != 0:
                self.key = None
                return False            # L2-style decay (grad += wd p) is torch's
            b1, b2 = g["betas"]
            hyper.group[gi] = _Group(float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                     float(decoupled_wd[gi]) if decoupled_wd is not None else 0.0)
        self.step_count += 1
        capi._chk(capi.lib().rslo_opt_adam_step(self.tensors_dev.data_ptr(), self.chunks_dev.data_ptr(), self.n_chunks,
                                                C.byref(hyper), float(self.step_count), capi._stream()),
                  "rslo_opt_adam_step")
        # a tensor with state but no gradient in this step keeps its count (torch skips it): it lags one more from now
        # on; the table is re-uploaded when it gets a gradient again (its gradient pointer changes then)
        skipped = self.has_state & (self.grad_ptrs == 0)
        if skipped.any():
            self.host["step_offset"][skipped] -= 1
        return True


def stepper_of(opt):
    """The AdamStepper of a torch optimizer (or of an OptimWrapper's inner one), created on first use; None if the
    optimizer is not a plain Adam over contiguous fp32 CUDA tensors."""
    inner = getattr(opt, "opt", opt)
    st = inner.__dict__.get("_rslo_stepper")
    if st is None:
        if not AdamStepper.supports(inner):
            return None
        st = inner.__dict__["_rslo_stepper"] = AdamStepper(inner)
    return st


def clip_grad_norm_(parameters, max_norm, optimizer=None):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) (train_hdf5.py:671); with `optimizer` (the OptimWrapper or
    torch optimizer that owns exactly these parameters) it runs on the optimizer's device tables in two launches."""
    st = stepper_of(optimizer) if optimizer is not None else None
    if st is not None:
        owned = sum(len(g["params"]) for g in st.opt.param_groups)
        params = parameters if isinstance(parameters, (list, tuple)) else list(parameters)
        if len(params) == owned:
            total = st.clip_grad_norm_(max_norm)
            if total is not None:
                return total
        parameters = params
    return torch.nn.utils.clip_grad_norm_(parameters, max_norm)
