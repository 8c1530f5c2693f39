"""Step-indexed lr / momentum schedules (reference: rslo/torchplus/train/learning_schedules_fastai.py:7-186).

A schedule is a list of phases (first_step, last_step, f) with f: progress in [0,1) -> value; at step s the LAST
phase whose first_step <= s wins, evaluated at (s - first) / (last - first).  `step(s)` writes optimizer.lr and
optimizer.mom (OptimWrapper setters: `mom` is Adam's beta1).  Phase starts are `int(fraction * total_step)`.
"""
from functools import partial

import numpy as np


def annealing_cos(start, end, pct):
    """Half-cosine from `start` (pct = 0) to `end` (pct = 1)."""
    return end + (start - end) / 2 * (np.cos(np.pi * pct) + 1)


def _constant(value, pct):
    return value


class LRSchedulerStep:
    def __init__(self, fai_optimizer, total_step, lr_phases, mom_phases):
        self.optimizer = fai_optimizer
        self.total_step = total_step
        self.lr_phases = self._table(lr_phases, total_step)
        self.mom_phases = self._table(mom_phases, total_step)
        assert self.lr_phases[0][0] == 0
        assert not self.mom_phases or self.mom_phases[0][0] == 0

    @staticmethod
    def _table(phases, total_step):
        phases = list(phases)
        table = []
        for i, (start, fn) in enumerate(phases):
            first = int(start * total_step)
            assert not table or table[-1][0] < first, "phase starts must increase"
            if isinstance(fn, str):
                fn = eval(fn)        # noqa: S307 -- lambda strings from the (trusted) training config, as in the reference
            last = int(phases[i + 1][0] * total_step) if i + 1 < len(phases) else total_step
            table.append((first, last, fn))
        return table

    @staticmethod
    def _value(table, step):
        val = None
        for first, last, fn in table:
            if step >= first:
                val = fn((step - first) / (last - first))
        return val

    def step(self, step):
        lr = self._value(self.lr_phases, step)
        if lr is not None:
            self.optimizer.lr = lr
        mom = self._value(self.mom_phases, step)
        if mom is not None:
            self.optimizer.mom = mom

    @property
    def learning_rate(self):
        return self.optimizer.lr


class OneCycle(LRSchedulerStep):
    """lr: lr_max/div -> lr_max over the first pct_start of training, then -> lr_max/div/1e4; momentum moms[0] ->
    moms[1] and back.  `lr_max` may be an array (one value per layer group)."""

    def __init__(self, fai_optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.lr_max, self.moms, self.div_factor, self.pct_start = lr_max, moms, div_factor, pct_start
        low = lr_max / div_factor
        lr_phases = ((0, partial(annealing_cos, low, lr_max)), (pct_start, partial(annealing_cos, lr_max, low / 1e4)))
        mom_phases = ((0, partial(annealing_cos, *moms)), (pct_start, partial(annealing_cos, *moms[::-1])))
        fai_optimizer.lr, fai_optimizer.mom = low, moms[0]
        super().__init__(fai_optimizer, total_step, lr_phases, mom_phases)


def _staircase(initial, decay_length, decay_factor, total_step, first_step):
    phases, step, value = [], first_step, initial
    while step <= total_step:
        phases.append((step / total_step, partial(_constant, value)))
        value = value * decay_factor
        step += int(decay_length * total_step)
    return phases


class ExponentialDecay(LRSchedulerStep):
    def __init__(self, fai_optimizer, total_step, initial_learning_rate, decay_length, decay_factor, staircase=True):
        assert 0 < decay_length < 1
        if staircase:
            phases = _staircase(initial_learning_rate, decay_length, decay_factor, total_step, 0)
        else:   # (the reference's smooth branch returns the bare factor, without the initial rate)
            phases = [(0, lambda p: pow(decay_factor, p / decay_length))]
        super().__init__(fai_optimizer, total_step, phases, [])


class ExponentialDecayWarmup(LRSchedulerStep):
    def __init__(self, fai_optimizer, total_step, initial_learning_rate, decay_length, decay_factor, div_factor=1,
                 pct_start=0, staircase=True):
        assert 0 < decay_length < 1
        phases = [(0, partial(annealing_cos, initial_learning_rate / div_factor, initial_learning_rate))]
        if staircase:
            phases += _staircase(initial_learning_rate, decay_length, decay_factor, total_step, pct_start * total_step)
        else:
            phases.append((pct_start, lambda p: pow(decay_factor, p / decay_length)))
        super().__init__(fai_optimizer, total_step, phases, [])


class ManualStepping(LRSchedulerStep):
    def __init__(self, fai_optimizer, total_step, boundaries, rates):
        assert all(0 < b < 1 for b in boundaries)
        assert len(boundaries) + 1 == len(rates)
        phases = [(start, partial(_constant, rate)) for start, rate in zip([0.0] + list(boundaries), rates)]
        super().__init__(fai_optimizer, total_step, phases, [])
