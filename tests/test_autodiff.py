"""Row a19, library path: nerfart_amd/autodiff.py (differentiable per-sample evaluation + compositing) against the
oracle's differentiable render, on the CPU, for sample depths taken from the oracle (the product's sampler is the
HIP kernel - covered by the GPU tests)."""
import numpy as np
import pytest
import torch

from conftest import scene_state, tt
from oracle import render as orender


def _grads(loss_terms, params):
    for p in params:
        p.grad = None
    for t in loss_terms:
        t()
    return [p.grad.clone() for p in params]


def test_volsdf_render_samples_matches_oracle_forward_and_backward(golden):
    from nerfart_amd import scene, autodiff
    from nerfart_amd import frameworks
    torch.manual_seed(0)
    model, _, _, _, _ = frameworks.get_model(scene.synthetic_config("VolSDF"))
    sd, rk = scene_state("VolSDF", 0.01)
    model.load_state_dict(sd)
    o, d = orender.get_rays(tt(golden["G9_c2w"]), tt(golden["G9_K"]), int(golden["G9_H"]), int(golden["G9_W"]))
    o, d = o.reshape(-1, 3)[:4], d.reshape(-1, 3)[:4]
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ex = orender.volsdf_render(sdg, o, d, near=rk["near"], far=rk["far"], obj_bounding_radius=rk["obj_bounding_radius"],
                               N_samples=128, max_upsample_steps=rk["max_upsample_steps"], differentiable=True)
    gvec = tt(golden["G11_gvec"])
    ex["rgb"].backward(gvec, retain_graph=True)
    nn_ = ex["implicit_nablas"].reshape(-1, 3).norm(dim=-1)
    (0.1 * torch.nn.functional.mse_loss(nn_, torch.ones_like(nn_))).backward()

    dn = torch.nn.functional.normalize(d, dim=-1)
    out = autodiff.volsdf_render_samples(model, o, dn, ex["d_vals"].detach())
    for k in ("rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_surface", "implicit_nablas", "sigma"):
        np.testing.assert_allclose(out[k].detach().numpy(), ex[k].detach().numpy(), atol=2e-5, rtol=2e-4, err_msg=k)
    model.zero_grad()
    out["rgb"].backward(gvec, retain_graph=True)
    n2 = out["implicit_nablas"].reshape(-1, 3).norm(dim=-1)
    (0.1 * torch.nn.functional.mse_loss(n2, torch.ones_like(n2))).backward()
    got = dict(model.named_parameters())
    assert len(got) == 43
    for name, p in got.items():
        ref = sdg[name].grad
        np.testing.assert_allclose(p.grad.numpy(), ref.numpy(), atol=1e-6 + 2e-4 * float(ref.abs().max()), rtol=1e-3, err_msg=name)
        # and the reference's own autograd (golden G11)
        np.testing.assert_allclose(float(p.grad.norm()), float(golden["G11_gradnorm_" + name]), rtol=2e-3, atol=1e-7, err_msg=name)


def test_neus_render_samples_matches_oracle_forward_and_backward(neus_state):
    from nerfart_amd import scene, autodiff, frameworks
    sd, rk = neus_state
    torch.manual_seed(0)
    model, _, _, _, _ = frameworks.get_model(scene.synthetic_config("NeuS"))
    model.load_state_dict(sd)
    c2w, K = scene.camera(6, 6)
    o, d = orender.get_rays(c2w, K, 6, 6)
    o, d = o.reshape(-1, 3)[10:14], d.reshape(-1, 3)[10:14]
    with torch.no_grad():
        ex = orender.neus_render(sd, o, d, obj_bounding_radius=rk.get("obj_bounding_radius", 1.0))
    dn = torch.nn.functional.normalize(d, dim=-1)
    d_all = ex["d_all"] if "d_all" in ex else ex["d_vals"]
    out = autodiff.neus_render_samples(model, o, dn, d_all)
    for k in ("rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_surface", "implicit_nablas", "radiance", "alpha"):
        np.testing.assert_allclose(out[k].detach().numpy(), ex[k].numpy(), atol=2e-5, rtol=2e-4, err_msg=k)
    model.zero_grad()
    out["rgb"].sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in model.named_parameters() if n.startswith("implicit_surface"))
