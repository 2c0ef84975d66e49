"""Pins the CPU oracle (oracle/) against golden vectors captured from the real reference
(tests/golden/make_golden.py).  Tolerances are fp32 round-off: the oracle issues the same torch ops
in (nearly) the same order, so most comparisons are exact or within a few ulp."""
import numpy as np
import pytest
import torch

from conftest import tt, scene_state, state_checksum
from oracle import nets, sampling, render


def close(a, b, atol=1e-6, rtol=1e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    same = (a == b) | (torch.isnan(a) & torch.isnan(b))          # covers +-inf
    err = torch.where(same, torch.zeros_like(a), (a - b).abs())
    assert torch.all(same | (err <= atol + rtol * b.abs())), f"max abs err {err.max().item():.3e}"


def test_G12_manifest_and_init(golden):
    for fw in ("VolSDF", "NeuS"):
        from nerfart_amd import scene, frameworks
        torch.manual_seed(0)
        model, _, _, _, _ = frameworks.get_model(scene.synthetic_config(fw))
        sd = model.state_dict()
        assert list(sd.keys()) == [str(k) for k in golden[f"G12_{fw}_keys"]]
        assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in golden[f"G12_{fw}_shapes"]]
        assert state_checksum(sd) == str(golden[f"G12_{fw}_init_sha256"])
        w4 = nets.folded_weight({k: v for k, v in sd.items()}, "implicit_surface.surface_fc_layers.4")
        close(w4[0], golden[f"G12_{fw}_fold_l4_row0"], 0, 0)


def test_G1_get_rays(golden):
    o, d = render.get_rays(tt(golden["G1_c2w"]), tt(golden["G1_K"]), 6, 5)
    close(o, golden["G1_rays_o"], 0, 0)
    close(d, golden["G1_rays_d"], 1e-6, 1e-6)
    assert np.array_equal(golden["G1_inds"], np.arange(30))


def test_G2_embed(golden):
    x = tt(golden["G2_x"])
    close(nets.embed(x, 6), golden["G2_e6"], 0, 0)
    close(nets.embed(x, 4), golden["G2_e4"], 0, 0)


def test_G3_G4_G5_networks(golden, volsdf_state):
    sd, _ = volsdf_state
    pts, v = tt(golden["G3_pts"]), tt(golden["G3_view"])
    sdf, feat = nets.surface_forward(sd, pts)
    close(sdf, golden["G3_sdf"], 1e-6, 1e-5)
    close(feat, golden["G3_feat"], 1e-6, 1e-5)
    s2, nab, _ = nets.surface_forward_with_nablas(sd, pts)
    close(nab, golden["G3_nabla"], 1e-5, 1e-5)
    close(nets.volsdf_forward_surface(sd, pts)[0], golden["G5_forward_surface"], 1e-6, 1e-5)
    rad, sdf_c, nab_c = nets.volsdf_forward(sd, pts, v)
    close(rad, golden["G5_radiance"], 1e-6, 1e-5)
    close(sdf_c, golden["G5_sdf"], 1e-6, 1e-5)
    close(nab_c, golden["G5_nabla"], 1e-5, 1e-5)
    close(nets.radiance_forward(sd, pts, v, tt(golden["G3_nabla"]), tt(golden["G3_feat"])), golden["G4_radiance"], 1e-6, 1e-5)
    assert (tt(golden["G5_sdf"]) < tt(golden["G3_sdf"])).any(), "fixture must exercise the sphere-background clamp"
    # independent forward-mode (fp64) evaluation agrees with reverse-mode autograd
    s64, n64, f64 = nets.surface_nablas_analytic(sd, pts)
    close(s64, golden["G3_sdf"], 2e-6, 1e-5)
    close(n64, golden["G3_nabla"], 2e-5, 1e-4)


def test_G6_sigma_and_error_bound(golden):
    d, s = tt(golden["G6_d"]), tt(golden["G6_s"])
    a, b = torch.tensor([100.0]), torch.tensor([0.01])
    close(sampling.sdf_to_sigma(s, a, b), golden["G6_sigma"], 0, 0)
    close(sampling.error_bound(d, s, a, b), golden["G6_bound_scalar"], 0, 1e-6)
    br = tt(golden["G6_beta_ray"])
    close(sampling.error_bound(d, s, 1.0 / br, br), golden["G6_bound_ray"], 0, 1e-6)
    nb = sampling.error_bound(tt(golden["G6_nan_d"]), tt(golden["G6_nan_s"]), torch.tensor([1e4]), torch.tensor([1e-4]))
    assert np.array_equal(np.isinf(nb.numpy()), np.isinf(golden["G6_nan_bound"]))
    assert np.isinf(golden["G6_nan_bound"]).any(), "fixture must exercise the NaN -> inf branch"


def test_G7_samplers(golden):
    bins, w, cdf = tt(golden["G7_bins"]), tt(golden["G7_w"]), tt(golden["G7_cdf"])
    close(sampling.sample_pdf(bins, w, 16), golden["G7_pdf16"], 0, 0)
    close(sampling.sample_pdf(bins, w, 66), golden["G7_pdf66"], 0, 0)
    close(sampling.sample_cdf(bins, cdf, 16), golden["G7_cdf16"], 0, 0)


@pytest.mark.parametrize("beta", [0.1, 0.01, 0.002])
def test_G8_fine_sample(golden, beta):
    sd, _ = scene_state("VolSDF", beta)
    H, W = int(golden["G9_H"]), int(golden["G9_W"])
    o, d = render.get_rays(tt(golden["G9_c2w"]), tt(golden["G9_K"]), H, W)
    dn = torch.nn.functional.normalize(d, dim=-1)
    alpha, bnet = nets.volsdf_ab(sd)
    t = torch.linspace(0, 1, 512)
    d_init = 0.0 * (1 - t) + 6.0 * torch.ones(H * W, 1) * t
    with torch.no_grad():
        d_fine, beta_map, usage = sampling.fine_sample(lambda x: nets.volsdf_forward_surface(sd, x)[0], d_init, o, dn, alpha, bnet,
                                                       6.0 * torch.ones(H * W, 1), eps=0.1, max_iter=6, max_bisection=10,
                                                       final_N_importance=64, N_up=512)
    tag = f"b{beta}"
    assert np.array_equal(usage.numpy(), golden[f"G8_{tag}_iter_usage"])
    close(beta_map, golden[f"G8_{tag}_beta_map"], 1e-7, 1e-5)
    close(d_fine, golden[f"G8_{tag}_d_fine"], 2e-5, 1e-5)
    if beta == 0.01:
        u = set(np.unique(golden[f"G8_{tag}_iter_usage"]).tolist())
        assert -1.0 in u and len(u) >= 3, "fixture must exercise several up-sampling rounds and the unconverged path"


@pytest.mark.parametrize("beta,ns", [(0.1, 128), (0.01, 32), (0.01, 128), (0.002, 128)])
def test_G9_volsdf_render(golden, beta, ns):
    sd, rk = scene_state("VolSDF", beta)
    o, d = render.get_rays(tt(golden["G9_c2w"]), tt(golden["G9_K"]), int(golden["G9_H"]), int(golden["G9_W"]))
    with torch.no_grad():
        out = render.volsdf_render(sd, o, d, near=rk["near"], far=rk["far"], obj_bounding_radius=rk["obj_bounding_radius"],
                                   N_samples=ns, max_upsample_steps=rk["max_upsample_steps"])
    tag = f"G9_b{beta}_n{ns}_"
    keys = [k[len(tag):] for k in golden if k.startswith(tag)]
    assert set(keys) == {"rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_surface", "implicit_nablas", "radiance",
                         "alpha", "p_i", "visibility_weights", "d_vals", "sigma", "beta_map", "iter_usage"}
    for k in keys:
        tol = dict(atol=3e-5, rtol=2e-4) if k in ("sigma", "implicit_nablas") else dict(atol=2e-5, rtol=1e-5)
        close(out[k], golden[tag + k], **tol)


def test_G10_neus(golden, neus_state):
    sd, rk = neus_state
    o, d = render.get_rays(tt(golden["G9_c2w"]), tt(golden["G9_K"]), int(golden["G9_H"]), int(golden["G9_W"]))
    dn = torch.nn.functional.normalize(d, dim=-1)
    near, far = render.near_far_from_sphere(o, dn, 1.0)
    close(near, golden["G10_near"], 0, 0); close(far, golden["G10_far"], 0, 0)
    cdf, alpha = render.sdf_to_alpha(tt(golden["G10_sdfp"]), torch.tensor([20.0]))
    close(cdf, golden["G10_cdf"], 0, 0); close(alpha, golden["G10_alpha"], 0, 0)
    close(render.alpha_to_w(alpha), golden["G10_w"], 0, 0)
    with torch.no_grad():
        out = render.neus_render(sd, o, d, obj_bounding_radius=rk["obj_bounding_radius"], N_upsample_iters=rk["N_upsample_iters"])
    tag = "G10_render_"
    keys = [k[len(tag):] for k in golden if k.startswith(tag)]
    assert set(keys) == {"rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_nablas", "implicit_surface", "radiance",
                         "alpha", "cdf", "visibility_weights", "d_final"}
    for k in keys:
        close(out[k], golden[tag + k], 3e-5, 2e-4)
    rad, sdf, nab = nets.neus_forward_radiance(sd, tt(golden["G10_pts"]), tt(golden["G10_view"])), None, None
    close(rad, golden["G10_radiance"], 1e-6, 1e-5)


def test_G11_backward_matches_reference(golden):
    """Trainer.forward pass 2 on 4 rays: rgb.backward(gvec) + 0.1 * eikonal MSE -> gradients of all 43 parameter
    tensors (reference autograd, double backward through the SDF net) vs the oracle's differentiable render."""
    from oracle import render as orender
    sd, rk = scene_state("VolSDF", 0.01)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o, d = orender.get_rays(tt(golden["G9_c2w"]), tt(golden["G9_K"]), int(golden["G9_H"]), int(golden["G9_W"]))
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    ex = orender.volsdf_render(sd, o[:4], d[:4], near=rk["near"], far=rk["far"], obj_bounding_radius=rk["obj_bounding_radius"],
                               N_samples=128, max_upsample_steps=rk["max_upsample_steps"], differentiable=True)
    ex["rgb"].backward(tt(golden["G11_gvec"]), retain_graph=True)
    nn_ = ex["implicit_nablas"].reshape(-1, 3).norm(dim=-1)
    (0.1 * torch.nn.functional.mse_loss(nn_, torch.ones_like(nn_))).backward()
    names = [k[len("G11_gradnorm_"):] for k in golden if k.startswith("G11_gradnorm_")]
    assert len(names) == 43
    for n in names:
        g = sd[n].grad
        assert g is not None, n
        ref_norm = float(golden["G11_gradnorm_" + n])
        np.testing.assert_allclose(float(g.norm()), ref_norm, rtol=2e-3, atol=1e-7, err_msg=n)
        head = golden["G11_gradhead_" + n]
        np.testing.assert_allclose(g.reshape(-1)[:head.size].numpy(), head, rtol=2e-2, atol=2e-3 * ref_norm / max(1.0, np.sqrt(g.numel())) + 1e-8, err_msg=n)


# ---- perturb=True (det=False samplers): the oracle takes the uniform numbers the reference drew ----------------------
def test_P1_samplers_with_given_uniforms(perturb_golden):
    pg = perturb_golden
    bins, w, cdf = tt(pg["P1_bins"]), tt(pg["P1_w"]), tt(pg["P1_cdf"])
    close(sampling.sample_pdf(bins, w, 16, det=False, u=tt(pg["P1_u_pdf"])), pg["P1_pdf16"], 0, 0)
    close(sampling.sample_cdf(bins, cdf, 16, det=False, u=tt(pg["P1_u_cdf"])), pg["P1_cdf16"], 0, 0)


def test_P2_P3_volsdf_perturb(perturb_golden):
    pg = perturb_golden
    sd, rk = scene_state("VolSDF", 0.01)
    assert state_checksum(sd) == str(pg["P2_state_sha256"])
    H, W = int(pg["P_H"]), int(pg["P_W"])
    o, d = render.get_rays(tt(pg["P_c2w"]), tt(pg["P_K"]), H, W)
    dn = torch.nn.functional.normalize(d, dim=-1)
    alpha, bnet = nets.volsdf_ab(sd)
    t = torch.linspace(0, 1, 512)
    d_init = 0.0 * (1 - t) + 6.0 * torch.ones(H * W, 1) * t
    with torch.no_grad():
        d_fine, beta_map, usage = sampling.fine_sample(lambda x: nets.volsdf_forward_surface(sd, x)[0], d_init, o, dn, alpha, bnet,
                                                       6.0 * torch.ones(H * W, 1), eps=0.1, max_iter=6, max_bisection=10,
                                                       final_N_importance=64, N_up=512, det=False, u_final=tt(pg["P2_u_final"]))
    assert np.array_equal(usage.numpy(), pg["P2_iter_usage"])
    close(beta_map, pg["P2_beta_map"], 1e-7, 1e-5)
    close(d_fine, pg["P2_d_fine"], 2e-5, 1e-5)
    assert not np.all(np.diff(pg["P2_d_fine"], axis=-1) >= 0), "fixture must hold unsorted (random) samples"
    with torch.no_grad():
        out = render.volsdf_render(sd, o, d, near=rk["near"], far=rk["far"], obj_bounding_radius=rk["obj_bounding_radius"],
                                   max_upsample_steps=rk["max_upsample_steps"], u_final=tt(pg["P3_u_final"]))
    for k in ("rgb", "depth_volume", "d_vals", "iter_usage", "beta_map"):
        close(out[k], pg["P3_" + k], 2e-5, 1e-5)


def test_P4_neus_perturb(perturb_golden, neus_state):
    pg = perturb_golden
    sd, rk = neus_state
    o, d = render.get_rays(tt(pg["P_c2w"]), tt(pg["P_K"]), int(pg["P_H"]), int(pg["P_W"]))
    with torch.no_grad():
        out = render.neus_render(sd, o, d, obj_bounding_radius=rk["obj_bounding_radius"], N_upsample_iters=rk["N_upsample_iters"],
                                 u_new=tt(pg["P4_u_new"]))
    for k in ("rgb", "depth_volume", "d_final", "implicit_surface"):
        close(out[k], pg["P4_" + k], 3e-5, 2e-4)


@pytest.mark.parametrize("algo", ["direct_use", "direct_more"])
def test_G10b_neus_direct_upsampling(neus_algos_golden, neus_state, algo):
    """neus.py:242-269: the two up-sampling algorithms besides 'official_solution' (YAML-reachable, neus.py:735) - every extras key at
    perturb=False, and rgb / depth / d_final at perturb=True with the reference's recorded draw."""
    ag = neus_algos_golden
    sd, rk = neus_state
    assert state_checksum(sd) == str(ag["A_state_sha256"])
    o, d = render.get_rays(tt(ag["A_c2w"]), tt(ag["A_K"]), int(ag["A_H"]), int(ag["A_W"]))
    with torch.no_grad():
        out = render.neus_render(sd, o, d, obj_bounding_radius=rk["obj_bounding_radius"], upsample_algo=algo)
    tag = f"A_{algo}_"
    for k in ("rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_nablas", "implicit_surface", "radiance", "alpha", "cdf",
              "visibility_weights", "d_final"):
        close(out[k], ag[tag + k], 3e-5, 2e-4)
    with torch.no_grad():
        out = render.neus_render(sd, o, d, obj_bounding_radius=rk["obj_bounding_radius"], upsample_algo=algo, u_new=tt(ag[tag + "perturb_u"]))
    for k in ("rgb", "depth_volume", "d_final", "mask_volume"):
        close(out[k], ag[tag + "perturb_" + k], 3e-5, 2e-4)


def test_the_two_passes_of_a_finetune_step_differ_only_in_the_final_inversion(volsdf_state):
    """What Trainer.share_algorithm1 rests on, from the reference itself and from the oracle.  (1) The reference's Trainer.forward at perturb=True
    (tests/golden/make_golden_finetune.py, keys FP_*): the rays of its pass 2 took exactly the rounds of its pass 1 (iter_usage equal on every
    ray) although the two passes drew different uniform numbers - Algorithm 1's rounds draw nothing (volsdf.py:159-285).  (2) The oracle: ONE
    fine_sample call that inverts every ray's final CDF at [u1 | u2] returns exactly what two calls with u1 and with u2 return."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "finetune_golden.npz"))
    np.testing.assert_array_equal(z["FP_VolSDF_iter_usage_pass1"], z["FP_VolSDF_iter_usage_pass2"])
    assert not np.array_equal(z["FP_VolSDF_u_pass1"], z["FP_VolSDF_u_pass2"])
    assert len(set(z["FP_VolSDF_iter_usage_pass1"].tolist())) >= 2
    sd, rk = volsdf_state
    o, d = render.get_rays(tt(z["F_c2w"]), tt(z["F_K"]), 8, 8)
    dn = torch.nn.functional.normalize(d, dim=-1)
    alpha, beta = nets.volsdf_ab(sd) if hasattr(nets, "volsdf_ab") else (None, None)
    if alpha is None:
        beta = torch.exp(sd["ln_beta"] * 10.0)
        alpha = 1.0 / beta
    t = torch.linspace(0, 1, 512).float()
    d_init = (0.0 * (1 - t) + 6.0 * t)[None, :].expand(64, 512)
    u1, u2 = tt(z["FP_VolSDF_u_pass1"]), tt(z["FP_VolSDF_u_pass2"])
    run = lambda u, n: sampling.fine_sample(lambda x: nets.volsdf_forward_surface(sd, x)[0], d_init, o, dn, alpha, beta, 6.0, eps=0.1, max_iter=6,
                                            max_bisection=10, final_N_importance=n, N_up=512, det=False, u_final=u)
    with torch.no_grad():
        d12, b12, us12 = run(torch.cat([u1, u2], 1), 128)
        d1, b1, us1 = run(u1, 64)
        d2, b2, us2 = run(u2, 64)
    assert torch.equal(us12, us1) and torch.equal(us12, us2) and torch.equal(b12, b1) and torch.equal(b12, b2)
    assert torch.equal(d12[:, :64], d1) and torch.equal(d12[:, 64:], d2)
    np.testing.assert_array_equal(us1.numpy(), z["FP_VolSDF_iter_usage_pass1"])


def test_oracle_views_fixture():
    """tests/golden/oracle_views_golden.npz (the oracle's rendering of 2,048 rays of 8 orbit views, written on the GPU box's host) IS the oracle's output:
    48 rays of every view re-rendered here.  Rays whose up-sampling takes the same rounds agree to 1e-4 (another CPU, another thread count, rays from the HIP get_rays);
    the rest are the never-converged / flipped rays any rounding moves - at most 2 of the 48."""
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    import make_oracle_views as mv
    z = np.load(os.path.join(here, "oracle_views_golden.npz"))
    assert tuple(z["poses"]) == mv.POSES and int(z["rays"]) == mv.N
    sd = mv.scene_sd()
    sub = torch.arange(5, mv.N, mv.N // 48)[:48]
    for p in mv.POSES:
        _, o, d = mv.view_rays(p, sub)
        ref = mv.render(sd, o, d)
        same = ref["iter_usage"].numpy() == z[f"pose{p}_iter_usage"][sub]
        err = np.abs(ref["rgb"].numpy() - z[f"pose{p}_rgb"][sub]).max(-1)
        assert same.sum() >= 46, (p, int(same.sum()))
        assert err[same & (ref["iter_usage"].numpy() >= 0)].max() < 1e-4, (p, float(err[same].max()))     # measured 3.3e-5 (8 threads here vs 32 on the GPU box; rays from the HIP get_rays there)
        assert err.max() < 1e-2
