"""A K-step trajectory of the reference's optimisation loop (train.py:232-248, models/base.py:486-575) against goldens of the REAL reference
(tests/golden/make_golden_trajectory.py): 5 Adam steps at 16 x 12 rays with the vangogh YAML's lr / exponential_step schedule, perturb=False -
the fine-tune branch (pixel MSE in place of the CLIP / VGG heads) for VolSDF and NeuS, and VolSDF's reconstruction branch on the reference's own
rays and eikonal points.  Held: the loss of every step, ||theta_5 - theta_0|| of every parameter tensor, and the image rendered from theta_5.

What it decides (VERDICT r4 missing 2 / next 5): the native backward's gradients carry ~4e-3 relative noise (single-term bf16 dumps); Adam divides
by sqrt(v), so a short run either absorbs that noise - or it does not, and the hi + lo dump option has to be built."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOSS_RTOL, RGB_ATOL, DTHETA_RTOL = 2e-3, 1e-3, 2e-2       # the tolerances VERDICT r4 "next 5" asked for: met where the reference meets them against itself
SLACK = 2.0                                               # ... elsewhere: within twice the reference's own deviation under one-ulp / 2^-17 weight noise (two seeds of a heavy-tailed quantity)


def _setup(fw, branch, z):
    from conftest import state_checksum
    from nerfart_amd import scene, optim
    from nerfart_amd.config import ConfigDict
    from nerfart_amd.trainer import Trainer
    tag = f"T_{fw}_{branch}_"
    model, rk_test, render_fn = scene.build_model(fw, seed=0, beta=0.01 if fw == "VolSDF" else None, device=DEV, precision="mixed")
    assert state_checksum({k: v.detach().cpu() for k, v in model.state_dict().items()}) == str(z[tag + "state_sha256"])
    args = ConfigDict({"training": ConfigDict({"is_finetune": branch == "finetune", "lr": 5.0e-4, "num_iters": 400, "w_eikonal": 0.1,
                                               "scheduler": ConfigDict({"type": "exponential_step", "min_factor": 0.5})}),
                       "finetune": ConfigDict({"w_eikonal": 0.1, "use_eikonal": True}), "data": ConfigDict({"N_rays": 96}),
                       "model": ConfigDict({"obj_bounding_radius": 3.0})})
    tr = Trainer(model, freeze_radiance=(fw == "NeuS" and branch == "finetune"))
    tr.render_fn = render_fn
    opt = optim.get_optimizer(args, model)
    sched = optim.get_scheduler(args, opt)
    return tag, model, rk_test, render_fn, args, tr, opt, sched


def _reference_bounds(case):
    """What the REFERENCE's own trajectory does under perturbations that are not errors (profiles/r06_trajectory_sensitivity.json,
    tools/trajectory_sensitivity.py: the real reference re-run with its initial SDF weights moved by one fp32 ulp / by 2^-17 relative, the size of the
    split-bf16 product error): the largest deviation from its unperturbed run over those runs, per statistic."""
    js = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_trajectory_sensitivity.json")))
    rows = [r for mode in ("ulp", "w2^-17") for r in js["cases"][case][mode]]
    return {"loss": max(max(r["loss_rel_err_per_step"]) for r in rows), "dtheta": max(r["worst_dtheta_norm_err"] for r in rows),
            "head": max(r["worst_dtheta_leading_entries_err"] for r in rows), "image": max(r["final_image_max_err"] for r in rows)}


def _finish(tag, z, model, theta0, render_fn, rk_test, losses, lrs, fw):
    from nerfart_amd import rend_util
    np.testing.assert_allclose(lrs, z[tag + "lr"], rtol=1e-12)
    rel = np.abs(np.array(losses) - z[tag + "loss"]) / np.abs(z[tag + "loss"])
    worst_d, worst_h, n, who_d, who_h = 0.0, 0.0, 0, "", ""
    for name, p in model.named_parameters():
        key = tag + "dnorm_" + name
        d = p.detach() - theta0[name]
        if key not in z.files:
            assert float(d.abs().max()) == 0.0, name
            continue
        n += 1
        gold = float(z[key])
        e = abs(float(d.norm()) - gold) / gold
        if e > worst_d:
            worst_d, who_d = e, name
        head = torch.from_numpy(z[tag + "dhead_" + name]).to(DEV)
        if float(head.norm()) > 1e-3 * gold:
            e = float((d.reshape(-1)[: head.numel()] - head).norm() / head.norm())
            if e > worst_h:
                worst_h, who_h = e, name
    H, W = int(z["T_H"]), int(z["T_W"])
    o, dd, _ = rend_util.get_rays(torch.from_numpy(z["T_c2w"])[None].to(DEV), torch.from_numpy(z["T_K"])[None].to(DEV), H, W)
    with torch.no_grad():
        rgb, _, _ = render_fn(o, dd, **({"require_nablas": True} if fw == "VolSDF" else {}), calc_normal=True, detailed_output=False, **rk_test)
    e_rgb = (rgb[0].cpu() - torch.from_numpy(z[tag + "final_rgb"])).abs().max(dim=-1).values
    bound = _reference_bounds(tag[2:-1])
    print(f"  {tag}: per-step loss error {['%.1e' % r for r in rel]}; worst ||dtheta|| error {worst_d:.2e} ({who_d}) over {n} tensors, worst leading-entries "
          f"error {worst_h:.2e} ({who_h}); image from theta_K: max {float(e_rgb.max()):.2e}, {int((e_rgb > 1e-3).sum())} of {e_rgb.numel()} rays past 1e-3; "
          f"the reference against itself (ulp / 2^-17 weight noise): {bound}")
    assert n == (28 if tag.startswith("T_NeuS") else 43)
    # the first two steps come before anything can amplify: hard
    assert float(rel[:2].max()) <= LOSS_RTOL, rel
    # from step 3 on the trajectory is the scene's, not the implementation's: Adam's sign-like first steps and Algorithm 1's branches make the
    # reference's OWN run move by `bound` when its weights change by an ulp - the native run has to stay inside what the reference does to itself
    assert float(rel.max()) <= max(LOSS_RTOL, SLACK * bound["loss"]), (rel, bound)
    assert worst_d <= max(DTHETA_RTOL, SLACK * bound["dtheta"]), (worst_d, who_d, bound)
    assert float(e_rgb.max()) <= max(RGB_ATOL, SLACK * bound["image"]), (float(e_rgb.max()), bound)
    return dict(loss=float(rel.max()), dtheta=worst_d, head=worst_h, image=float(e_rgb.max()))


@pytest.mark.parametrize("fw", ["VolSDF", "NeuS"])
def test_finetune_trajectory_matches_the_reference_loop(fw):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_golden.npz"))
    tag, model, rk_test, render_fn, args, tr, opt, sched = _setup(fw, "finetune", z)
    tr.style_loss = lambda pred, gt: ((pred - gt) ** 2).mean()
    rk = json.loads(str(z[tag + "render_kwargs"]))
    model_input = {"intrinsics": torch.from_numpy(z["T_K"])[None], "c2w": torch.from_numpy(z["T_c2w"])[None]}
    ground_truth = {"rgb": torch.from_numpy(z["T_target"])[None]}
    theta0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses, lrs = [], []
    for it in range(int(z["T_steps"])):
        lrs.append(opt.param_groups[0]["lr"])
        ret = tr(args, torch.tensor([0]), model_input, ground_truth, rk, it, optimizer=opt)       # train.py:232 (the backward is inside)
        losses.append(float(ret["losses"]))
        opt.step()                                                                                  # train.py:247-248
        sched.step(it)
    _finish(tag, z, model, theta0, render_fn, rk_test, losses, lrs, fw)


def test_reconstruction_trajectory_matches_the_reference_loop():
    from nerfart_amd import rend_util
    fw = "VolSDF"
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_golden.npz"))
    tag, model, rk_test, render_fn, args, tr, opt, sched = _setup(fw, "recon", z)
    for p in model.parameters():
        p.requires_grad_(True)
    rk = json.loads(str(z[tag + "render_kwargs"]))
    H, W = rk.pop("H"), rk.pop("W")
    o, d, _ = rend_util.get_rays(torch.from_numpy(z["T_c2w"])[None].to(DEV), torch.from_numpy(z["T_K"])[None].to(DEV), H, W)
    target = torch.from_numpy(z["T_target"]).to(DEV)
    theta0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses, lrs = [], []
    for it in range(int(z["T_steps"])):
        lrs.append(opt.param_groups[0]["lr"])
        sel = torch.from_numpy(z[tag + "select_inds"][it]).to(DEV)                 # the rays get_rays' randint drew in the reference's step
        pts = torch.from_numpy(z[tag + "eikonal_points"][it]).to(DEV)              # ... and its uniform eikonal points (volsdf.py:799-801)
        opt.zero_grad()
        out = tr.reconstruction_step(render_fn, o[0, sel], d[0, sel], target[sel], eikonal_points=pts, w_eikonal=float(z[tag + "w_eikonal"]), **rk)
        losses.append(out["total"])
        opt.step()
        sched.step(it)
    _finish(tag, z, model, theta0, render_fn, rk_test, losses, lrs, fw)
