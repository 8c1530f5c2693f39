"""fastai-style optimizer wrapper used by the training driver (reference: rslo/torchplus/train/fastai_optim.py:14-25,
119-304).

Contract kept:
  * every *layer group* owns TWO torch param groups: [non-BatchNorm leaves, BatchNorm leaves] (only
    nn.BatchNorm{1,2,3}d count as BatchNorm, fastai_optim.py:12 -- apex/torch SyncBatchNorm lands in the first half);
  * hyper-parameters (`lr`, `mom`, `beta`, `wd`) are per layer group and are written into both of its param groups;
    `mom` is Adam's beta1 (or SGD momentum), `beta` is Adam's beta2 (or RMSprop alpha);
  * `true_wd`: decoupled weight decay, p *= 1 - wd * lr BEFORE the inner step, inner weight_decay forced to 0;
  * state_dict / load_state_dict / param_groups are the inner optimizer's (checkpoint format unchanged).

MI355X form: on contiguous fp32 CUDA parameters under a plain Adam the decay and the update of ALL tensors run in one
launch (rslo_amd.optim.AdamStepper over csrc/optim.hip; state stays in the inner optimizer's dict).  Otherwise the
decoupled decay is one multi-tensor launch per param group (torch._foreach_mul_) instead of one launch per parameter
tensor (213 tensors), and the inner optimizer may be built with fused=True by the builder.
"""
from collections.abc import Iterable

import torch
from torch import nn

bn_types = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


def listify(p=None, q=None):
    """`p` as a list with the length of `q` (an int, a sized object, or None = len(p)); singletons broadcast."""
    if p is None:
        p = []
    elif isinstance(p, str) or not isinstance(p, Iterable):
        p = [p]
    p = list(p)
    n = q if isinstance(q, int) else (len(p) if q is None else len(q))
    if len(p) == 1:
        p = p * n
    assert len(p) == n, "List len mismatch (%d vs %d)" % (len(p), n)
    return p


def trainable_params(m):
    return [p for p in m.parameters() if p.requires_grad]


def split_bn_bias(layer_groups):
    """[group, ...] -> [non-BN children, BN children, ...] (two nn.Sequential per group)."""
    out = []
    for group in layer_groups:
        plain, bn = [], []
        for child in group.children():
            (bn if isinstance(child, bn_types) else plain).append(child)
        out += [nn.Sequential(*plain), nn.Sequential(*bn)]
    return out


class OptimWrapper:
    """Wraps a torch optimizer whose param groups come in (non-BN, BN) pairs."""

    def __init__(self, opt, wd, true_wd=False, bn_wd=True):
        self.opt, self.true_wd, self.bn_wd = opt, true_wd, bn_wd
        self.opt_keys = [k for k in self.opt.param_groups[0].keys() if k != "params"]
        self.read_defaults()
        self.wd = wd
        self.param_segs = []

    # ------------------------------------------------------------------ construction
    @classmethod
    def create(cls, opt_func, lr, layer_groups, **kwargs):
        """layer_groups: either [module] or a list of such one-element lists (the driver passes 4 of them:
        VFE, middle, odom head, losses -- optimizer_builder.py:49-66)."""
        if len(layer_groups) == 1:
            split = split_bn_bias(layer_groups)
        else:
            split = []
            for lg in layer_groups:
                split += split_bn_bias(lg)
        opt = cls(opt_func([{"params": trainable_params(g), "lr": 0} for g in split]), **kwargs)
        opt.lr, opt.opt_func = listify(lr, layer_groups), opt_func
        return opt

    def new(self, layer_groups):
        return self.create(getattr(self, "opt_func", self.opt.__class__), self.lr, layer_groups, wd=self.wd,
                           true_wd=self.true_wd, bn_wd=self.bn_wd)

    def __repr__(self):
        return "OptimWrapper over %r.\nTrue weight decay: %s" % (self.opt, self.true_wd)

    # ------------------------------------------------------------------ optimizer protocol
    def _pairs(self):
        g = self.opt.param_groups
        return zip(g[::2], g[1::2])

    @torch.no_grad()
    def step(self):
        if self._hip_step():
            return
        if self.true_wd:
            for lr, wd, (plain, bn) in zip(self._lr, self._wd, self._pairs()):
                for grp in ((plain, bn) if self.bn_wd else (plain,)):
                    ps = [p for p in grp["params"]]
                    if ps:
                        torch._foreach_mul_(ps, 1 - wd * lr)
            self.set_val("weight_decay", listify(0, self._wd))
        self.opt.step()

    def _hip_step(self):
        """Decay + Adam over all tensors in one launch (rslo_amd.optim / csrc/optim.hip) when the inner optimizer is a
        plain Adam over contiguous fp32 CUDA tensors; False -> the torch formulation below runs."""
        from rslo_amd import optim as hip_optim
        st = hip_optim.stepper_of(self.opt)
        if st is None:
            return False
        wd = None
        if self.true_wd:
            self.set_val("weight_decay", listify(0, self._wd))
            wd = []
            for w in self._wd:
                wd += [w, w if self.bn_wd else 0.0]
        return st.step(wd)

    def zero_grad(self, set_to_none=True):
        self.opt.zero_grad(set_to_none=set_to_none)

    def __getstate__(self):
        return self.opt.__getstate__()

    def __setstate__(self, state):
        return self.opt.__setstate__(state)

    def state_dict(self):
        return self.opt.state_dict()

    def load_state_dict(self, state_dict):
        return self.opt.load_state_dict(state_dict)

    def add_param_group(self, param_group):
        return self.opt.add_param_group(param_group)

    def clear(self):
        sd = self.state_dict()
        sd["state"] = {}
        self.load_state_dict(sd)

    @property
    def param_groups(self):
        return self.opt.param_groups

    @property
    def defaults(self):
        return self.opt.defaults

    @property
    def state(self):
        return self.opt.state

    # ------------------------------------------------------------------ hyper-parameters
    @property
    def lr(self):
        return self._lr[-1]

    @lr.setter
    def lr(self, val):
        self._lr = self.set_val("lr", listify(val, self._lr))

    @property
    def mom(self):
        return self._mom[-1]

    @mom.setter
    def mom(self, val):
        if "momentum" in self.opt_keys:
            self.set_val("momentum", listify(val, self._mom))
        elif "betas" in self.opt_keys:
            self.set_val("betas", (listify(val, self._mom), self._beta))
        self._mom = listify(val, self._mom)

    @property
    def beta(self):
        return None if self._beta is None else self._beta[-1]

    @beta.setter
    def beta(self, val):
        if val is None:
            return
        if "betas" in self.opt_keys:
            self.set_val("betas", (self._mom, listify(val, self._beta)))
        elif "alpha" in self.opt_keys:
            self.set_val("alpha", listify(val, self._beta))
        self._beta = listify(val, self._beta)

    @property
    def wd(self):
        return self._wd[-1]

    @wd.setter
    def wd(self, val):
        if not self.true_wd:
            self.set_val("weight_decay", listify(val, self._wd), bn_groups=self.bn_wd)
        self._wd = listify(val, self._wd)

    def read_defaults(self):
        self._beta = None
        if "lr" in self.opt_keys:
            self._lr = self.read_val("lr")
        if "momentum" in self.opt_keys:
            self._mom = self.read_val("momentum")
        if "alpha" in self.opt_keys:
            self._beta = self.read_val("alpha")
        if "betas" in self.opt_keys:
            self._mom, self._beta = self.read_val("betas")
        if "weight_decay" in self.opt_keys:
            self._wd = self.read_val("weight_decay")

    def set_val(self, key, val, bn_groups=True):
        """Writes one value per layer group into its (non-BN[, BN]) param groups; a tuple of lists zips into
        per-group tuples (betas)."""
        if isinstance(val, tuple):
            val = [(a, b) for a, b in zip(*val)]
        for v, (plain, bn) in zip(val, self._pairs()):
            plain[key] = v
            if bn_groups:
                bn[key] = v
        return val

    def read_val(self, key):
        val = [g[key] for g in self.opt.param_groups[::2]]
        if isinstance(val[0], tuple):
            val = [o[0] for o in val], [o[1] for o in val]
        return val


class FastAIMixedOptim(OptimWrapper):
    """fp16 master-weight variant (fastai_optim.py:307-354): the driver always builds with mixed=False
    (train_hdf5.py:408-411, optimizer_builder.py:112), so it is outside the path."""

    @classmethod
    def create(cls, *args, **kwargs):
        raise NotImplementedError("FastAIMixedOptim is not used by the RSLO training path (mixed=False)")
