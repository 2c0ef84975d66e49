"""Optimiser and learning-rate schedules of the training loop (the contract of the reference's
models/base.py:486-575 as train.py:114, :158, :247-248 uses it): stock `torch.optim.Adam` with per-name learning rates, and
stock `LambdaLR` / `MultiStepLR` over two factor curves.  Pinned against the reference's functions
(tests/golden/make_golden_optim.py -> tests/test_optim.py)."""
import numbers

import numpy as np
from torch.optim import Adam
from torch.optim.lr_scheduler import LambdaLR, MultiStepLR


def _lr_groups(model, default_lr, table):
    """Adam parameter groups for {'<direct parameter or sub-module of model>': lr}: group 0 = everything the table does not
    name at default_lr, then one group per table entry in table order (the order the reference's checkpoints store)."""
    params = dict(model.named_parameters())
    named_groups, taken = [], set()
    for key, lr in table.items():
        members = [n for n in params if n == key or n.startswith(key + ".")]
        if not members or "." in key:
            raise RuntimeError("wrong lr key:", key)
        taken.update(members)
        named_groups.append({"params": [params[n] for n in members], "lr": lr})
    rest = {"params": [p for n, p in params.items() if n not in taken], "lr": default_lr}
    return [rest] + named_groups


def get_optimizer(args, model):
    """`args.training.lr`: a number, or {'default': lr, '<parameter or sub-module name>': lr, ...}.  The 'default' entry is
    removed from the config's dict (callers that save the config afterwards see that in the reference as well)."""
    lr = args.training.lr
    if isinstance(lr, numbers.Number):
        return Adam(model.parameters(), lr=lr)
    if not isinstance(lr, dict):
        raise NotImplementedError(f"training.lr of type {type(lr).__name__}")
    default_lr = lr.pop("default")
    return Adam(_lr_groups(model, default_lr, lr), lr=default_lr)


def warmup_cosine_factor(step, total_steps, warmup_steps, min_factor=0.1):
    """step / warmup_steps up to warmup_steps, then half a cosine from 1 down to min_factor at total_steps."""
    if step < warmup_steps:
        return step / warmup_steps
    return (np.cos(np.pi * ((step - warmup_steps) / (total_steps - warmup_steps))) + 1.0) * 0.5 * (1 - min_factor) + min_factor


def exponential_factor(step, total_steps, min_factor=0.1):
    """min_factor ** clip(step / total_steps, 0, 1), evaluated as exp(t * log(min_factor))."""
    return np.exp(np.clip(step / total_steps, 0, 1) * np.log(min_factor))


def _curve(fn, **fixed):
    assert 0 <= fixed["min_factor"] < 1
    return lambda step: fn(step, **fixed)


# the reference's names for the two curve factories (configs and notebooks import them)
def CosineAnnealWarmUpSchedulerLambda(total_steps, warmup_steps, min_factor=0.1):
    return _curve(warmup_cosine_factor, total_steps=total_steps, warmup_steps=warmup_steps, min_factor=min_factor)


def ExponentialSchedulerLambda(total_steps, min_factor=0.1):
    return _curve(exponential_factor, total_steps=total_steps, min_factor=min_factor)


def get_scheduler(args, optimizer, last_epoch=-1):
    """`args.training.scheduler.type` in {multistep, warmupcosine, exponential_step}.  A missing min_factor is written back
    into the config as 0.1; 'exponential_step' starts from step 0 whatever last_epoch says (both as in the reference)."""
    sc, total = args.training.scheduler, args.training.get("num_iters", None)
    kind = sc.type
    if kind == "multistep":
        return MultiStepLR(optimizer, sc.milestones, gamma=sc.gamma, last_epoch=last_epoch)
    if kind == "warmupcosine":
        return LambdaLR(optimizer, CosineAnnealWarmUpSchedulerLambda(total, sc.warmup_steps, sc.setdefault("min_factor", 0.1)),
                        last_epoch=last_epoch)
    if kind == "exponential_step":
        return LambdaLR(optimizer, ExponentialSchedulerLambda(total, sc.setdefault("min_factor", 0.1)))
    raise NotImplementedError(f"scheduler type {kind!r}")
