"""LR scheduler from the optimizer message (reference: rslo/builder/lr_scheduler_builder.py:22-115)."""
import numpy as np

from torchplus.train import learning_schedules_fastai as lsf


def build(optimizer_config, optimizer, total_step):
    kind = optimizer_config.WhichOneof("optimizer")
    if kind not in ("rms_prop_optimizer", "momentum_optimizer", "adam_optimizer"):
        raise ValueError("Optimizer %s not supported." % kind)
    return _create_learning_rate_scheduler(getattr(optimizer_config, kind).learning_rate, optimizer, total_step)


def _create_learning_rate_scheduler(learning_rate_config, optimizer, total_step):
    kind = learning_rate_config.WhichOneof("learning_rate")
    if kind == "multi_phase":
        phases = learning_rate_config.multi_phase.phases
        return lsf.LRSchedulerStep(optimizer, total_step, [(p.start, p.lambda_func) for p in phases],
                                   [(p.start, p.momentum_lambda_func) for p in phases])
    if kind == "one_cycle":
        cfg = learning_rate_config.one_cycle
        if len(cfg.lr_maxs) > 1:      # one peak per layer group
            assert len(cfg.lr_maxs) == 4
            lr_max = np.array(list(cfg.lr_maxs))
        else:
            lr_max = cfg.lr_max
        return lsf.OneCycle(optimizer, total_step, lr_max, list(cfg.moms), cfg.div_factor, cfg.pct_start)
    if kind == "exponential_decay":
        cfg = learning_rate_config.exponential_decay
        return lsf.ExponentialDecay(optimizer, total_step, cfg.initial_learning_rate, cfg.decay_length,
                                    cfg.decay_factor, cfg.staircase)
    if kind == "exponential_decay_warmup":
        cfg = learning_rate_config.exponential_decay_warmup
        return lsf.ExponentialDecayWarmup(optimizer, total_step, cfg.initial_learning_rate, cfg.decay_length,
                                          cfg.decay_factor, cfg.div_factor, cfg.pct_start, cfg.staircase)
    if kind == "manual_stepping":
        cfg = learning_rate_config.manual_stepping
        return lsf.ManualStepping(optimizer, total_step, list(cfg.boundaries), list(cfg.rates))
    raise ValueError("Learning_rate %s not supported." % kind)
