"""Golden vectors for the FINE-TUNE branch of the reference Trainer.forward (volsdf.py:719-783, neus.py:520-576), from the
REAL reference on CPU:

    python tests/golden/make_golden_finetune.py          -> tests/golden/finetune_golden.npz

`trainer.forward(args, indices, model_input, ground_truth, render_kwargs_train, it, optimizer=opt)` as train.py:232 calls
it, perturb=False, an 8 x 8 image (one 64-ray patch), with `calc_style_loss` replaced by a pixel MSE (the CLIP / VGG
checkpoints are not available offline; the style heads are pinned separately, tests/test_clip.py, tests/test_vgg.py) -
everything else is the reference's: pass 1, d loss / d rgb, pass 2 with rgb.backward(gradient) and the eikonal backward,
the NeuS radiance net frozen by `fix_module`-style requires_grad (neus.py:455-456).  Stored: the style loss, the image of
pass 1, and the gradient norm + leading 32 entries of every parameter that received one.

Keys `FP_*`: the same call at perturb=True, the reference's default (volsdf.py:982, neus.py:742) - pass 2 then draws NEW samples for
the gradient of a loss evaluated on pass 1's.  torch.rand is recorded while the reference runs; stored per pass: the uniform numbers
as one [R, 64] table, the rendered rgb (and iter_usage), then the loss and the gradients as above.
"""
import os
import sys
import types

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
    g = torch.Generator().manual_seed(78)
    target = torch.rand(1, H * W, 3, generator=g) * 0.3 + 0.5
    out.update(F_c2w=c2w, F_K=K, F_target=target[0])
    for fw, yaml_name, beta in (("VolSDF", "volsdf_fangzhou_nature.yaml", 0.01), ("NeuS", "neus_fangzhou.yaml", None)):
        cfg = io_util.load_yaml(os.path.join(mg.REF, "configs", yaml_name))
        cfg.device_ids = ["cpu"]
        cfg.training.is_finetune = False              # build the Trainer WITHOUT the CLIP / VGG heads ...
        torch.manual_seed(0)
        model, trainer, rk_train, rk_test, _ = ref_get_model(cfg, [480, 270])
        cfg.training.is_finetune = True               # ... and run its fine-tune branch with a pixel loss in their place
        cfg.finetune = {"use_eikonal": True, "w_eikonal": 0.1, "w_perceptual": 2.0, "target_text": "painting"}
        trainer.neg_texts = []
        trainer.calc_style_loss = types.MethodType(lambda self, rgb, rgb_gt, args, H=480: ((rgb - rgb_gt) ** 2).mean(), trainer)
        if fw == "NeuS":                              # what Trainer.__init__ does when is_finetune (neus.py:455-456)
            for p in model.radiance_net.parameters():
                p.requires_grad_(False)
        torch.manual_seed(0)
        mine, _, _, _, _ = frameworks.get_model(scene.synthetic_config(fw))
        sd = scene.perturb_state(mine.state_dict(), beta=beta, seed=1)
        model.load_state_dict(sd)
        out[f"F_{fw}_state_sha256"] = np.array(mg.state_checksum(sd))
        rk = dict(rk_train)
        rk["perturb"] = False
        rk["H"], rk["W"] = H, W
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0)
        model_input = {"intrinsics": K[None], "c2w": c2w[None]}
        ground_truth = {"rgb": target}
        orig_to, orig_cuda = torch.Tensor.to, torch.Tensor.cuda      # neus.py:505 hard-codes device = "cuda"
        torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else orig_to(self, *a, **k)
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            with np.errstate(all="ignore"):
                ret = trainer.forward(cfg, torch.tensor([0]), model_input, ground_truth, rk, 0, optimizer=opt)
        finally:
            torch.Tensor.to, torch.Tensor.cuda = orig_to, orig_cuda
        tag = f"F_{fw}_"
        import json
        out[tag + "render_kwargs"] = np.array(json.dumps({k: v for k, v in rk.items() if isinstance(v, (int, float, bool, str))}))
        out[tag + "loss"] = ret["losses"].detach()
        n = 0
        for name, p in model.named_parameters():
            if p.grad is None:
                continue
            n += 1
            out[tag + "gradnorm_" + name] = p.grad.norm()
            out[tag + "gradhead_" + name] = p.grad.reshape(-1)[:32].clone()
        print(fw, "style loss", float(ret["losses"]), "parameters with gradients:", n)

        # ---- the same call at the reference's DEFAULT perturb=True (volsdf.py:982, neus.py:742): both passes call the renderer with
        # render_kwargs_train, so pass 2 draws NEW uniform numbers and back-propagates pass 1's d loss / d rgb through its own samples.
        # torch.rand is recorded per renderer call and re-assembled into one [R, 64] table per pass (the form the C ABI takes).
        from make_golden_perturb import RandRecorder
        rk["perturb"] = True
        R = H * W
        calls = []
        inner = trainer.renderer
        orig_forward = inner.forward

        def recording_forward(*a, **k):
            n0 = len(rr.draws)
            res = orig_forward(*a, **k)
            calls.append((n0, len(rr.draws), res[0].detach().clone(), res[2]))
            return res
        inner.forward = recording_forward
        torch.Tensor.to = lambda self, *a, **k: self if (a and isinstance(a[0], str) and a[0].startswith("cuda")) else orig_to(self, *a, **k)
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            with RandRecorder() as rr, np.errstate(all="ignore"):
                torch.manual_seed(5)
                ret = trainer.forward(cfg, torch.tensor([0]), model_input, ground_truth, rk, 0, optimizer=opt)
        finally:
            torch.Tensor.to, torch.Tensor.cuda = orig_to, orig_cuda
            del inner.forward
        assert len(calls) == 2, len(calls)                      # pass 1 (one 64-ray chunk) and pass 2 (one 64-ray patch)
        tag = f"FP_{fw}_"
        for pno, (n0, n1, rgb_call, ex) in enumerate(calls, start=1):
            draws = rr.draws[n0:n1]
            if fw == "VolSDF":                                  # one draw per converged subset, rounds ascending, the never-converged rest last
                usage = ex["iter_usage"][0].detach()
                order = [k for k in sorted(set(usage.tolist()) - {-1.0})] + ([-1.0] if (usage == -1).any() else [])
                assert len(order) == len(draws), (order, [tuple(x.shape) for x in draws])
                u = torch.zeros(R, 64)
                for k, dr in zip(order, draws):
                    mk = usage == k
                    assert int(mk.sum()) == dr.reshape(-1, 64).shape[0]
                    u[mk] = dr.reshape(-1, 64)
                out[tag + f"iter_usage_pass{pno}"] = usage
            else:                                               # NeuS: one [R, 16] draw per up-sampling round
                assert len(draws) == 4 and all(x.reshape(-1, 16).shape[0] == R for x in draws), [tuple(x.shape) for x in draws]
                u = torch.cat([x.reshape(R, 16) for x in draws], dim=-1)
            out[tag + f"u_pass{pno}"] = u
            out[tag + f"rgb_pass{pno}"] = rgb_call[0]
        assert not torch.equal(out[tag + "u_pass1"], out[tag + "u_pass2"])
        out[tag + "render_kwargs"] = np.array(json.dumps({k: v for k, v in rk.items() if isinstance(v, (int, float, bool, str))}))
        out[tag + "loss"] = ret["losses"].detach()
        n = 0
        for name, p in model.named_parameters():
            if p.grad is None:
                continue
            n += 1
            out[tag + "gradnorm_" + name] = p.grad.norm()
            out[tag + "gradhead_" + name] = p.grad.reshape(-1)[:32].clone()
        print(fw, "perturb=True: style loss", float(ret["losses"]), "parameters with gradients:", n,
              "max |rgb pass 1 - rgb pass 2|", float((calls[0][2] - calls[1][2]).abs().max()))
    np.savez_compressed(os.path.join(HERE, "finetune_golden.npz"), **mg.t2n(out))
    print("wrote finetune_golden.npz")


if __name__ == "__main__":
    main()
