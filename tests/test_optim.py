"""get_optimizer / get_scheduler against the reference's (tests/golden/make_golden_optim.py)."""
import copy
import os
import warnings

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim_golden.npz")


@pytest.fixture(scope="module")
def og():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


def test_schedule_lambdas(og):
    from nerfart_amd import optim as no
    steps = og["steps"]
    np.testing.assert_allclose([no.CosineAnnealWarmUpSchedulerLambda(300000, 5000, 0.1)(int(s)) for s in steps], og["cos_300k_5k"], rtol=0, atol=0)
    np.testing.assert_allclose([no.CosineAnnealWarmUpSchedulerLambda(400, 100, 0.0)(int(s)) for s in steps[:9]], og["cos_400_100_0"], rtol=0, atol=0)
    np.testing.assert_allclose([no.ExponentialSchedulerLambda(400, 0.5)(int(s)) for s in steps], og["exp_400_05"], rtol=0, atol=0)


def test_optimizer_groups_and_trajectories(og):
    from nerfart_amd import frameworks, scene, optim as no
    from nerfart_amd.config import ConfigDict
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    model, _, _, _, _ = frameworks.get_model(scene.synthetic_config("VolSDF"))
    args = ConfigDict({"training": ConfigDict({"lr": {"default": 5.0e-4, "ln_beta": 1.0e-3, "radiance_net": 2.0e-4}})})
    opt = no.get_optimizer(copy.deepcopy(args), model)
    assert isinstance(opt, torch.optim.Adam)
    np.testing.assert_array_equal([g["lr"] for g in opt.param_groups], og["dict_group_lr"])
    np.testing.assert_array_equal([sum(p.numel() for p in g["params"]) for g in opt.param_groups], og["dict_group_numel"])
    with pytest.raises(RuntimeError):
        no.get_optimizer(ConfigDict({"training": ConfigDict({"lr": {"default": 1e-3, "no_such_module": 1e-4}})}), model)
    for stype, extra in (("exponential_step", {"min_factor": 0.5}), ("warmupcosine", {"warmup_steps": 3}),
                         ("multistep", {"milestones": [2, 5], "gamma": 0.5})):
        a = ConfigDict({"training": ConfigDict({"lr": 5.0e-4, "num_iters": 8, "scheduler": ConfigDict(dict(type=stype, **extra))})})
        o = no.get_optimizer(a, model)
        sch = no.get_scheduler(a, o, last_epoch=-1)
        traj = [o.param_groups[0]["lr"]]
        for it in range(8):
            o.step()
            sch.step(it)                                      # train.py:248
            traj.append(o.param_groups[0]["lr"])
        np.testing.assert_allclose(traj, og[f"traj_{stype}"], rtol=1e-15, atol=0, err_msg=stype)
