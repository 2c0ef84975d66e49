"""Golden vectors for get_optimizer / get_scheduler (reference models/base.py:486-575), from the REAL reference:

    python tests/golden/make_golden_optim.py          -> tests/golden/optim_golden.npz

Stored: the learning-rate factor of each schedule at a list of steps, the per-group learning rates and parameter
counts of the optimiser built from an lr dictionary, and the lr trajectory of optimiser + scheduler over a few
train.py-style `scheduler.step(it)` calls.
"""
import copy
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    warnings.filterwarnings("ignore")
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    from models import base as ref_base
    from addict import Dict
    import nerfart_amd  # noqa: F401
    from nerfart_amd import scene, frameworks
    out = {}
    steps = np.array([0, 1, 10, 99, 100, 101, 250, 399, 400, 401, 1000, 5000, 299999, 300000])
    out["steps"] = steps
    out["cos_300k_5k"] = np.array([ref_base.CosineAnnealWarmUpSchedulerLambda(300000, 5000, 0.1)(int(s)) for s in steps])
    out["cos_400_100_0"] = np.array([ref_base.CosineAnnealWarmUpSchedulerLambda(400, 100, 0.0)(int(s)) for s in steps[:9]])
    out["exp_400_05"] = np.array([ref_base.ExponentialSchedulerLambda(400, 0.5)(int(s)) for s in steps])
    torch.manual_seed(0)
    model, _, _, _, _ = frameworks.get_model(scene.synthetic_config("VolSDF"))
    args = Dict()
    args.training.lr = {"default": 5.0e-4, "ln_beta": 1.0e-3, "radiance_net": 2.0e-4}
    opt = ref_base.get_optimizer(copy.deepcopy(args), model)
    out["dict_group_lr"] = np.array([g["lr"] for g in opt.param_groups])
    out["dict_group_numel"] = np.array([sum(p.numel() for p in g["params"]) for g in opt.param_groups])
    for stype, extra in (("exponential_step", {"min_factor": 0.5}), ("warmupcosine", {"warmup_steps": 3}),
                         ("multistep", {"milestones": [2, 5], "gamma": 0.5})):
        a = Dict()
        a.training.lr = 5.0e-4
        a.training.num_iters = 8
        a.training.scheduler = dict(type=stype, **extra)
        o = ref_base.get_optimizer(a, model)
        sch = ref_base.get_scheduler(a, o, last_epoch=-1)
        traj = [o.param_groups[0]["lr"]]
        for it in range(8):
            o.step()
            sch.step(it)                     # train.py:248
            traj.append(o.param_groups[0]["lr"])
        out[f"traj_{stype}"] = np.array(traj)
    np.savez_compressed(os.path.join(HERE, "optim_golden.npz"), **out)
    print({k: v for k, v in out.items() if k.startswith("traj") or k.startswith("dict")})


if __name__ == "__main__":
    main()
