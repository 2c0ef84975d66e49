"""Golden vectors for the reconstruction branch of the reference Trainer.forward (volsdf.py:784-824, neus.py:578-617), from
the REAL reference on CPU:

    python tests/golden/make_golden_recon.py          -> tests/golden/recon_golden.npz

`trainer.forward(args, indices, model_input, ground_truth, render_kwargs_train, it)` is called as train.py:232 does, with
perturb=False and N_rays = 12 random rays of an 8 x 8 image; `losses['total'].backward()` as train.py:242.  Stored: the
selected ray indices, the uniform eikonal points the VolSDF branch drew (recorded by wrapping Tensor.uniform_), the
target pixels / mask, the losses, and the gradient norm + leading 32 entries of every parameter.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    from utils import io_util
    from models.frameworks import get_model as ref_get_model
    import nerfart_amd  # noqa: F401
    from nerfart_amd import scene, frameworks
    torch.set_num_threads(8)
    out = {}
    H = W = 8
    c2w, K = scene.camera(H, W)
    g = torch.Generator().manual_seed(77)
    target = torch.rand(1, H * W, 3, generator=g)
    mask = torch.rand(1, H * W, generator=g) > 0.35
    out.update(R_c2w=c2w, R_K=K, R_target=target[0], R_mask=mask[0])
    for fw, yaml_name, beta in (("VolSDF", "volsdf_fangzhou_nature.yaml", 0.01), ("NeuS", "neus_fangzhou.yaml", None)):
        cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", yaml_name))
        cfg.device_ids = ["cpu"]                     # Trainer.device = device_ids[0] (volsdf.py:634): keeps every .to(device) on the CPU
        cfg.training.is_finetune = False
        cfg.data.N_rays = 12
        if fw == "NeuS":
            cfg.training.with_mask = True
            cfg.training.w_mask = 0.3
        torch.manual_seed(0)
        model, trainer, rk_train, rk_test, _ = ref_get_model(cfg, [480, 270])
        torch.manual_seed(0)
        mine, _, _, _, _ = frameworks.get_model(scene.synthetic_config(fw))
        sd = scene.perturb_state(mine.state_dict(), beta=beta, seed=1)
        model.load_state_dict(sd)
        out[f"R_{fw}_state_sha256"] = np.array(mg.state_checksum(sd))
        rk = dict(rk_train)
        rk["perturb"] = False
        rk["H"], rk["W"] = H, W
        drawn = []
        orig = torch.Tensor.uniform_

        def rec(self, *a, **k):
            r = orig(self, *a, **k)
            drawn.append(r.detach().clone())
            return r
        trainer.neg_texts = []                      # (forward builds the fine-tune prompt list even when reconstructing, volsdf.py:697)
        model_input = {"intrinsics": K[None], "c2w": c2w[None], "object_mask": mask}
        ground_truth = {"rgb": target}
        model.zero_grad()
        torch.manual_seed(5)
        torch.Tensor.uniform_ = rec
        # neus.py:505 hard-codes device = "cuda": on this GPU-less box every .to("cuda") / .cuda() is made a no-op while it runs
        orig_to, orig_cuda = torch.Tensor.to, torch.Tensor.cuda
        torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else orig_to(self, *a, **k)
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            ret = trainer.forward(cfg, torch.tensor([0]), model_input, ground_truth, rk, 0)
        finally:
            torch.Tensor.uniform_ = orig
            torch.Tensor.to, torch.Tensor.cuda = orig_to, orig_cuda
        losses = ret["losses"]
        for k, v in losses.items():
            losses[k] = torch.mean(v)
        losses["total"].backward()
        tag = f"R_{fw}_"
        import json
        out[tag + "render_kwargs"] = np.array(json.dumps({k: v for k, v in rk.items() if isinstance(v, (int, float, bool, str))}))
        out[tag + "w_eikonal"] = np.array(float(cfg.training.w_eikonal))
        out[tag + "select_inds"] = ret["extras"]["select_inds"][0]
        for k, v in losses.items():
            out[tag + k] = v.detach()
        if fw == "VolSDF":
            assert len(drawn) == 1, [tuple(x.shape) for x in drawn]
            out[tag + "eikonal_points"] = drawn[0].reshape(-1, 3)
        for name, p in model.named_parameters():
            if p.grad is None:
                continue
            out[tag + "gradnorm_" + name] = p.grad.norm()
            out[tag + "gradhead_" + name] = p.grad.reshape(-1)[:32].clone()
        print(fw, {k: float(v) for k, v in losses.items()}, "rays", out[tag + "select_inds"].tolist())
    np.savez_compressed(os.path.join(HERE, "recon_golden.npz"), **mg.t2n(out))
    print("wrote recon_golden.npz")


if __name__ == "__main__":
    main()
