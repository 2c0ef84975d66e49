"""Golden vectors for a K-step TRAJECTORY of the reference's optimisation loop (train.py:232-248: trainer.forward -> [zero_grad, total.backward]
-> optimizer.step -> scheduler.step(it); models/base.py:486-575: Adam from get_optimizer, the `exponential_step` LambdaLR), from the REAL
reference on CPU:

    python tests/golden/make_golden_trajectory.py          -> tests/golden/trajectory_golden.npz

5 Adam steps on a 16 x 12 image (192 rays), perturb=False, lr / scheduler / num_iters of configs/volsdf_fangzhou_vangogh.yaml (5e-4,
exponential_step, min_factor 0.5, 400 iterations):
  * fine-tune branch (VolSDF and NeuS; calc_style_loss replaced by a pixel MSE as in make_golden_finetune.py): all 192 rays per step;
  * reconstruction branch (VolSDF): data.N_rays = 96 rays per step drawn by get_rays' torch.randint, eikonal points by Tensor.uniform_ - both
    recorded per step so that a run elsewhere can take the same rays and points.
Stored per case: the learning rate and loss(es) of every step, ||theta_K - theta_0|| and the leading 16 entries of theta_K - theta_0 for every
parameter tensor, and the image rendered from theta_K (render_kwargs_test, all 192 rays).  What this pins that the one-step goldens do not:
whether the ~4e-3 relative noise of the native gradients (single-term bf16 dumps) moves a short Adam run.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

K_STEPS = 5
SGD_LR_FILE = os.path.join(HERE, "trajectory_sgd_lr.json")        # per case: the SGD learning rate tools/trajectory_sensitivity.py settled on


def main(opt_kind="adam", H=16, W=12, out_name="trajectory_golden.npz"):
    """opt_kind 'adam': the reference's own optimiser (get_optimizer) - trajectory_golden.npz.  'sgd' (`--sgd`): the SAME loop with
    torch.optim.SGD(lr) under the same exponential_step schedule, 24 x 16 rays, lr per case from trajectory_sgd_lr.json - a WELL-CONDITIONED
    trajectory: Adam's first steps are +-lr per entry whatever the gradient's size, so its 5-step run amplifies one-ulp
    noise to 0.08 - 0.73 in the image and can hold nothing after step 2; SGD's update is linear in the gradient, the reference's own one-ulp
    deviation stays below 1e-3 (tools/trajectory_sensitivity.py, profiles/r08_trajectory_sensitivity_sgd.json) and the GPU test holds all 5 steps hard."""
    sgd_lr = json.load(open(SGD_LR_FILE)) if opt_kind == "sgd" else None
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    from utils import io_util, rend_util
    from models.frameworks import get_model as ref_get_model
    from models.base import get_optimizer, get_scheduler
    import nerfart_amd  # noqa: F401
    from nerfart_amd import scene, frameworks
    torch.set_num_threads(8)
    out = {}
    c2w, K = scene.camera(H, W)
    g = torch.Generator().manual_seed(79)
    target = torch.rand(1, H * W, 3, generator=g) * 0.3 + 0.5
    out.update(T_c2w=c2w, T_K=K, T_target=target[0], T_H=np.array(H), T_W=np.array(W), T_steps=np.array(K_STEPS))
    train_cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", "volsdf_fangzhou_vangogh.yaml")).training
    cases = (("VolSDF", "volsdf_fangzhou_nature.yaml", 0.01, "finetune"), ("NeuS", "neus_fangzhou.yaml", None, "finetune"),
             ("VolSDF", "volsdf_fangzhou_nature.yaml", 0.01, "recon"))
    for fw, yaml_name, beta, branch in cases:
        cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", yaml_name))
        cfg.device_ids = ["cpu"]
        cfg.training.is_finetune = False                      # build the Trainer without the CLIP / VGG heads (make_golden_finetune.py)
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
        model.load_state_dict(sd)
        tag = f"T_{fw}_{branch}_"
        out[tag + "state_sha256"] = np.array(mg.state_checksum(sd))
        theta0 = {n: p.detach().clone() for n, p in model.named_parameters()}
        if opt_kind == "sgd":
            cfg.training.lr = float(sgd_lr[f"{fw}_{branch}"])
            optimizer = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=cfg.training.lr)
        else:
            optimizer = get_optimizer(cfg, model)
        scheduler = get_scheduler(cfg, optimizer)
        rk = dict(rk_train)
        rk["perturb"] = False
        rk["H"], rk["W"] = H, W
        model_input = {"intrinsics": K[None], "c2w": c2w[None]}
        ground_truth = {"rgb": target}
        drawn = []
        orig_uniform = torch.Tensor.uniform_

        def rec(self, *a, **k):
            r = orig_uniform(self, *a, **k)
            drawn.append(r.detach().clone())
            return r
        orig_to, orig_cuda = torch.Tensor.to, torch.Tensor.cuda
        torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else orig_to(self, *a, **k)
        torch.Tensor.cuda = lambda self, *a, **k: self
        lrs, losses, sel, eik_pts, parts = [], [], [], [], {"loss_img": [], "loss_eikonal": []}
        torch.manual_seed(9)
        try:
            for it in range(K_STEPS):
                lrs.append(optimizer.param_groups[0]["lr"])
                drawn.clear()
                torch.Tensor.uniform_ = rec
                try:
                    with np.errstate(all="ignore"):
                        ret = trainer.forward(cfg, torch.tensor([0]), model_input, ground_truth, rk, it, optimizer=optimizer)
                finally:
                    torch.Tensor.uniform_ = orig_uniform
                ls = ret["losses"]
                if branch == "recon":                          # train.py:236-242
                    for k, v in ls.items():
                        ls[k] = torch.mean(v)
                    optimizer.zero_grad()
                    ls["total"].backward()
                    losses.append(float(ls["total"]))
                    for k in parts:
                        parts[k].append(float(ls[k]))
                    sel.append(ret["extras"]["select_inds"][0].clone())
                    assert len(drawn) == 1, [tuple(x.shape) for x in drawn]
                    eik_pts.append(drawn[0].reshape(-1, 3))
                else:
                    losses.append(float(ls))
                optimizer.step()
                scheduler.step(it)
        finally:
            torch.Tensor.to, torch.Tensor.cuda = orig_to, orig_cuda
        out[tag + "lr"] = np.array(lrs, dtype=np.float64)
        out[tag + "loss"] = np.array(losses, dtype=np.float64)
        if branch == "recon":
            out[tag + "select_inds"] = torch.stack(sel)
            out[tag + "eikonal_points"] = torch.stack(eik_pts)
            out[tag + "w_eikonal"] = np.array(float(cfg.training.w_eikonal))
            for k, v in parts.items():
                out[tag + k] = np.array(v, dtype=np.float64)
        out[tag + "render_kwargs"] = np.array(json.dumps({k: v for k, v in rk.items() if isinstance(v, (int, float, bool, str))}))
        n = 0
        for name, p in model.named_parameters():
            d = p.detach() - theta0[name]
            if float(d.abs().max()) == 0.0:
                continue
            n += 1
            out[tag + "dnorm_" + name] = d.norm()
            out[tag + "dhead_" + name] = d.reshape(-1)[:16].clone()
        ro, rd, _ = rend_util.get_rays(c2w[None], K[None], H, W)
        with torch.no_grad():
            rgb, _, _ = ref_render(ro, rd, **({"require_nablas": True} if fw == "VolSDF" else {}), calc_normal=True, detailed_output=False, **rk_test)
        out[tag + "final_rgb"] = rgb[0]
        print(tag, "lr", [f"{x:.3e}" for x in lrs], "loss", [round(x, 6) for x in losses], "tensors that moved:", n)
    np.savez_compressed(os.path.join(HERE, out_name), **mg.t2n(out))
    print("wrote", out_name, len(out), "arrays")


if __name__ == "__main__":
    if "--sgd" in sys.argv:
        main("sgd", 24, 16, "trajectory_sgd_golden.npz")
    else:
        main()
