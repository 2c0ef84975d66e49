#!/usr/bin/env python
"""How far does the REFERENCE's own 5-step Adam trajectory (tests/golden/make_golden_trajectory.py: train.py:232-248 at lr 5e-4) move under
perturbations that are not errors?  Runs the real reference (/root/reference, build container only) several times per case:

    ulp      the initial SDF weights multiplied by (1 + 2^-23 U(-1, 1)): what another GEMM summation order / libm does
    w2^-17   ... by (1 + 2^-17 U(-1, 1)): the size of the split-bf16 kernels' product error (3 bf16 MFMAs per product), as a weight perturbation
    g1e-4    every parameter gradient multiplied elementwise by (1 + 1e-4 N(0, 1)) before optimizer.step()
    g4e-3    ... by (1 + 4e-3 N(0, 1)): the measured relative error of the native backward's gradients (single-term bf16 dumps)
    ga4e-3   every gradient entry PLUS 4e-3 x rms(gradient tensor) x N(0, 1): the same error as additive noise (entries far below the rms lose
             their sign - and Adam gives every entry a step of the same size); ga3e-4: the norm error the one-step goldens measure for NeuS

and reports, against the unperturbed run (= the golden): per-step loss error, worst ||theta_5 - theta_0|| error and worst leading-entries error
over the parameter tensors, max error of the image rendered from theta_5.  Adam's first steps are sign-like (m / sqrt(v) = +-1): an element whose
gradient is smaller than the noise gets a +-lr update of random sign under ANY of these - this measures how much of the native path's deviation
from the golden trajectory is that, and not the size of its gradient error.

    python tools/trajectory_sensitivity.py          -> profiles/r06_trajectory_sensitivity.json
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402

K_STEPS = 5
MODES = tuple(os.environ.get("TRAJ_MODES", "ulp,w2^-17,g1e-4,g4e-3").split(","))
OUT_NAME = os.environ.get("TRAJ_OUT", "r06_trajectory_sensitivity.json")
# TRAJ_OPT=sgd (round 6): the same loop with torch.optim.SGD under the same schedule at TRAJ_HW (24x16) rays; TRAJ_SGD_LR: a JSON object
# {case: lr}, default = tests/golden/trajectory_sgd_lr.json.  `TRAJ_OPT=sgd TRAJ_FIND_LR=1`: per case, start from the lr whose 5-step
# ||theta_5 - theta_0|| equals the Adam golden's and halve it until the reference's OWN one-ulp deviation of the final image is <= 1e-3; writes the json.
OPT = os.environ.get("TRAJ_OPT", "adam")
HW = tuple(int(v) for v in os.environ.get("TRAJ_HW", "16x12" if OPT == "adam" else "24x16").split("x"))
SGD_LR_FILE = os.path.join(ROOT, "tests", "golden", "trajectory_sgd_lr.json")


def run(fw, yaml_name, beta, branch, mode, seed, recorded=None, sgd_lr=None):
    from utils import io_util, rend_util
    from models.frameworks import get_model as ref_get_model
    from models.base import get_optimizer, get_scheduler
    from nerfart_amd import scene, frameworks
    H, W = HW
    c2w, K = scene.camera(H, W)
    g = torch.Generator().manual_seed(79)
    target = torch.rand(1, H * W, 3, generator=g) * 0.3 + 0.5
    train_cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", "volsdf_fangzhou_vangogh.yaml")).training
    cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", yaml_name))
    cfg.device_ids = ["cpu"]
    cfg.training.is_finetune = False
    cfg.training.lr = float(train_cfg.lr)
    cfg.training.num_iters = int(train_cfg.num_iters)
    cfg.training.scheduler = {"type": train_cfg.scheduler.type, "min_factor": float(train_cfg.scheduler.min_factor)}
    cfg.data.N_rays = 96
    torch.manual_seed(0)
    model, trainer, rk_train, rk_test, ref_render = ref_get_model(cfg, [480, 270])
    if branch == "finetune":
        cfg.training.is_finetune = True
        cfg.finetune = {"use_eikonal": True, "w_eikonal": 0.1, "w_perceptual": 2.0, "target_text": "painting"}
        trainer.calc_style_loss = types.MethodType(lambda self, rgb, rgb_gt, args, H=480: ((rgb - rgb_gt) ** 2).mean(), trainer)
        if fw == "NeuS":
            for p in model.radiance_net.parameters():
                p.requires_grad_(False)
    trainer.neg_texts = []
    torch.manual_seed(0)
    mine, _, _, _, _ = frameworks.get_model(scene.synthetic_config(fw))
    sd = scene.perturb_state(mine.state_dict(), beta=beta, seed=1)
    gen = torch.Generator().manual_seed(1000 + seed)
    if mode in ("ulp", "w2^-17"):
        amp = 2.0 ** -23 if mode == "ulp" else 2.0 ** -17
        for k in sd:
            if k.startswith("implicit_surface") and k.endswith(("weight_v", "weight_g")):
                sd[k] = sd[k] * (1.0 + amp * (2 * torch.rand(sd[k].shape, generator=gen) - 1))
    model.load_state_dict(sd)
    theta0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    if OPT == "sgd":
        cfg.training.lr = float(sgd_lr)
        optimizer = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=cfg.training.lr)
    else:
        optimizer = get_optimizer(cfg, model)
    scheduler = get_scheduler(cfg, optimizer)
    rk = dict(rk_train); rk["perturb"] = False; rk["H"], rk["W"] = H, W
    model_input = {"intrinsics": K[None], "c2w": c2w[None]}
    ground_truth = {"rgb": target}
    orig_to, orig_cuda = torch.Tensor.to, torch.Tensor.cuda
    torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else orig_to(self, *a, **k)
    torch.Tensor.cuda = lambda self, *a, **k: self
    losses = []
    torch.manual_seed(9)                                    # the same rays / eikonal points as the golden run in every variant
    try:
        for it in range(K_STEPS):
            with np.errstate(all="ignore"):
                ret = trainer.forward(cfg, torch.tensor([0]), model_input, ground_truth, rk, it, optimizer=optimizer)
            ls = ret["losses"]
            if branch == "recon":
                for k, v in ls.items():
                    ls[k] = torch.mean(v)
                optimizer.zero_grad()
                ls["total"].backward()
                losses.append(float(ls["total"]))
            else:
                losses.append(float(ls))
            if mode.startswith("ga"):                      # additive: every entry moves by amp x the tensor's rms gradient (what rounding the
                amp = float(mode[2:])                      # operands of a weight-gradient reduction does: small entries lose their sign)
                for p in model.parameters():
                    if p.grad is not None:
                        p.grad.add_(amp * p.grad.pow(2).mean().sqrt() * torch.randn(p.grad.shape, generator=gen))
            elif mode.startswith("g"):
                amp = float(mode[1:])
                for p in model.parameters():
                    if p.grad is not None:
                        p.grad.mul_(1.0 + amp * torch.randn(p.grad.shape, generator=gen))
            optimizer.step()
            scheduler.step(it)
    finally:
        torch.Tensor.to, torch.Tensor.cuda = orig_to, orig_cuda
    ro, rd, _ = rend_util.get_rays(c2w[None], K[None], H, W)
    with torch.no_grad():
        rgb, _, _ = ref_render(ro, rd, **({"require_nablas": True} if fw == "VolSDF" else {}), calc_normal=True, detailed_output=False, **rk_test)
    dth = {n: (p.detach() - theta0[n]) for n, p in model.named_parameters()}
    return dict(loss=np.array(losses), rgb=rgb[0].clone(), dtheta=dth)


def find_sgd_lr(cases):
    """Per case: the SGD learning rate whose 5-step ||theta_5 - theta_0|| (all tensors) equals the Adam golden's, halved until the reference's OWN
    deviation of the final image under one-ulp weight noise (two seeds) is <= 1e-3 - the condition under which a 5-step test can be held hard."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "trajectory_golden.npz"))
    lrs, log = (json.load(open(SGD_LR_FILE)) if os.path.exists(SGD_LR_FILE) else {}), {}
    for fw, yaml_name, beta, branch in cases:
        case = f"{fw}_{branch}"
        adam = float(np.sqrt(sum(float(z[k]) ** 2 for k in z.files if k.startswith(f"T_{case}_dnorm_"))))
        probe_lr = 1e-3
        probe = run(fw, yaml_name, beta, branch, "none", 0, sgd_lr=probe_lr)
        sgd = float(np.sqrt(sum(float(d.norm()) ** 2 for d in probe["dtheta"].values())))
        lr = float(f"{probe_lr * adam / sgd:.2e}")
        log[case] = {"adam_dtheta_norm": adam, "sgd_dtheta_norm_at_1e-3": sgd, "tried": []}
        while True:
            base = run(fw, yaml_name, beta, branch, "none", 0, sgd_lr=lr)
            dn = float(np.sqrt(sum(float(d.norm()) ** 2 for d in base["dtheta"].values())))
            loss = [float(x) for x in base["loss"]]
            rec = {"lr": lr, "dtheta_norm": dn, "loss": loss, "final_image_std": float(base["rgb"].std())}
            log[case]["tried"].append(rec)
            # a trajectory worth holding: the loss goes down at every step (no overshoot - matching Adam's ||dtheta|| concentrates the whole move on the
            # few entries with large gradients and diverges) and the image is still an image
            stable = all(b < a for a, b in zip(loss, loss[1:])) and rec["final_image_std"] > 1e-2 and np.isfinite(loss).all()
            if stable:
                devs = []
                for seed in range(2):
                    r = run(fw, yaml_name, beta, branch, "ulp", seed, sgd_lr=lr)
                    devs.append(float((r["rgb"] - base["rgb"]).abs().max()))
                rec["one_ulp_image_dev"] = [float(f"{d:.2e}") for d in devs]
            print(case, rec, flush=True)
            if (stable and max(devs) <= 1e-3) or lr < 1e-7:
                break
            lr = float(f"{lr / (2 if stable else 4):.2e}")
        lrs[case] = lr
        json.dump(lrs, open(SGD_LR_FILE, "w"), indent=1, sort_keys=True)
    json.dump(log, open(os.path.join(ROOT, "profiles", "r08_trajectory_sgd_lr_search.json"), "w"), indent=1)
    print("wrote", SGD_LR_FILE, lrs)


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    torch.set_num_threads(8)
    out = {"what": __doc__.split("\n\n")[0], "lr": 5e-4, "steps": K_STEPS, "rays": 192, "cases": {}}
    cases = (("VolSDF", "volsdf_fangzhou_nature.yaml", 0.01, "finetune"), ("NeuS", "neus_fangzhou.yaml", None, "finetune"),
             ("VolSDF", "volsdf_fangzhou_nature.yaml", 0.01, "recon"))
    only = os.environ.get("TRAJ_CASES")
    if only:
        cases = tuple(c for c in cases if f"{c[0]}_{c[3]}" in only.split(","))
    if OPT == "sgd":
        out.update(optimizer="torch.optim.SGD under the same exponential_step schedule", rays=HW[0] * HW[1])
        if os.environ.get("TRAJ_FIND_LR"):
            return find_sgd_lr(cases)
        sgd_lrs = json.loads(os.environ["TRAJ_SGD_LR"]) if os.environ.get("TRAJ_SGD_LR") else json.load(open(SGD_LR_FILE))
        out["lr"] = sgd_lrs
    for fw, yaml_name, beta, branch in cases:
        kw = {"sgd_lr": sgd_lrs[f"{fw}_{branch}"]} if OPT == "sgd" else {}
        base = run(fw, yaml_name, beta, branch, "none", 0, **kw)
        res = {}
        for mode in MODES:
            rows = []
            for seed in range(2):
                r = run(fw, yaml_name, beta, branch, mode, seed, **kw)
                worst_d = worst_h = 0.0
                for n, d0 in base["dtheta"].items():
                    if float(d0.abs().max()) == 0.0:
                        continue
                    worst_d = max(worst_d, abs(float(r["dtheta"][n].norm()) - float(d0.norm())) / float(d0.norm()))
                    h0 = d0.reshape(-1)[:16]
                    if float(h0.norm()) > 1e-3 * float(d0.norm()):
                        worst_h = max(worst_h, float((r["dtheta"][n].reshape(-1)[:16] - h0).norm() / h0.norm()))
                e = (r["rgb"] - base["rgb"]).abs().max(dim=-1).values
                rows.append({"loss_rel_err_per_step": [float(f"{x:.2e}") for x in np.abs(r["loss"] - base["loss"]) / np.abs(base["loss"])],
                             "worst_dtheta_norm_err": float(f"{worst_d:.2e}"), "worst_dtheta_leading_entries_err": float(f"{worst_h:.2e}"),
                             "final_image_max_err": float(f"{float(e.max()):.2e}"), "final_image_rays_past_1e-3": int((e > 1e-3).sum())})
                print(fw, branch, mode, seed, rows[-1], flush=True)
            res[mode] = rows
        out["cases"][f"{fw}_{branch}"] = res
    json.dump(out, open(os.path.join(ROOT, "profiles", OUT_NAME), "w"), indent=1)
    print("wrote profiles/" + OUT_NAME)


if __name__ == "__main__":
    main()
