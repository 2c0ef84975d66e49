"""Optimiser and learning-rate schedules of the training loop (reference models/base.py:486-575,
used by train.py:114, :158, :247-248): Adam with an optional per-sub-module / per-parameter lr dictionary, and the three
schedules the configs name (multistep, warmupcosine, exponential_step).  Pinned against the reference's functions
(tests/golden/make_golden_optim.py -> tests/test_optim.py)."""
import numbers

import numpy as np
from torch import optim


def get_optimizer(args, model):
    """args.training.lr: a number, or {'default': lr, '<parameter or sub-module name>': lr, ...} (base.py:486-521)."""
    lr = args.training.lr
    if isinstance(lr, numbers.Number):
        return optim.Adam(model.parameters(), lr=lr)
    if not isinstance(lr, dict):
        raise NotImplementedError
    default_lr = lr.pop("default")                      # (the reference pops it too: the config dict loses the key)
    groups, chosen = [], []
    for name, value in lr.items():
        if name in model._parameters.keys():
            chosen.append(name)
            groups.append({"params": getattr(model, name), "lr": value})
        elif name in model._modules.keys():
            sub = getattr(model, name)
            chosen.extend("{}.{}".format(name, pn) for pn, _ in sub.named_parameters())
            groups.append({"params": sub.parameters(), "lr": value})
        else:
            raise RuntimeError("wrong lr key:", name)
    rest = [p for n, p in model.named_parameters() if n not in chosen]
    groups.insert(0, {"params": rest, "lr": default_lr})
    return optim.Adam(params=groups, lr=default_lr)


def CosineAnnealWarmUpSchedulerLambda(total_steps, warmup_steps, min_factor=0.1):
    """Linear warm-up to 1, then half a cosine down to min_factor at total_steps (base.py:524-535)."""
    assert 0 <= min_factor < 1

    def lambda_fn(epoch):
        if epoch < warmup_steps:
            return epoch / warmup_steps
        phase = (epoch - warmup_steps) / (total_steps - warmup_steps)
        return (np.cos(np.pi * phase) + 1.0) * 0.5 * (1 - min_factor) + min_factor
    return lambda_fn


def ExponentialSchedulerLambda(total_steps, min_factor=0.1):
    """min_factor ** clip(step / total_steps, 0, 1)  (base.py:538-544)."""
    assert 0 <= min_factor < 1

    def lambda_fn(epoch):
        return np.exp(np.clip(epoch / total_steps, 0, 1) * np.log(min_factor))
    return lambda_fn


def get_scheduler(args, optimizer, last_epoch=-1):
    """base.py:547-575.  As there, 'exponential_step' ignores last_epoch."""
    sc = args.training.scheduler
    if sc.type == "multistep":
        return optim.lr_scheduler.MultiStepLR(optimizer, sc.milestones, gamma=sc.gamma, last_epoch=last_epoch)
    if sc.type == "warmupcosine":
        fn = CosineAnnealWarmUpSchedulerLambda(total_steps=args.training.num_iters, warmup_steps=sc.warmup_steps,
                                               min_factor=sc.setdefault("min_factor", 0.1))
        return optim.lr_scheduler.LambdaLR(optimizer, fn, last_epoch=last_epoch)
    if sc.type == "exponential_step":
        fn = ExponentialSchedulerLambda(total_steps=args.training.num_iters, min_factor=sc.setdefault("min_factor", 0.1))
        return optim.lr_scheduler.LambdaLR(optimizer, fn)
    raise NotImplementedError
