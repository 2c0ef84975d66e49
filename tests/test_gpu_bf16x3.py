"""GPU parity of the split-bf16 ("bf16x3") matrix-core path: same C ABI, precision = 1.

Tolerances: the split evaluates every product to ~2^-16 relative (3 bf16 MFMAs, fp32 accumulate), measured on
the CPU model as 2e-5 max / 3e-6 mean sdf error.  Point queries: sdf 1e-4, nabla 1e-3, radiance 5e-4.
Rendered pixels: every ray whose up-sampling took the same number of rounds as the reference within 1e-3 (the
north_star bound); >= 97 % of all rays take the same number of rounds (the bound check `max B > eps` sits on
a threshold: the CPU model of this arithmetic flips 0.2 % of rays, fp32 itself 0.07 %).
"""
import numpy as np
import pytest
import torch

from conftest import scene_state, tt
from test_gpu_parity import close, report, DEV

pytestmark = pytest.mark.gpu


SAMPLERS = [None, "fp16x2"]          # Algorithm 1 at the model's precision (the fused renderer) / on the 2-MFMA kernels (the mixed mode)


def _model(fw="VolSDF", beta=0.01, sampler=None):
    from nerfart_amd import scene
    model, rk, fn = scene.build_model(fw, seed=0, beta=beta, device=DEV, precision="bf16x3")
    if fw == "VolSDF":
        model.set_sampler_precision(sampler)
    return model, rk, fn


@pytest.fixture(scope="module")
def pts():
    g = torch.Generator().manual_seed(12)
    p = torch.rand(1500, 3, generator=g) * 6 - 3
    p[:500] *= 0.35
    v = torch.nn.functional.normalize(torch.randn(1500, 3, generator=g), dim=-1)
    return p, v


@pytest.mark.parametrize("M", [1, 31, 128, 129, 1500])
def test_sdf_fwd_bf16x3(pts, M):
    from oracle import nets
    model, _, _ = _model()
    sd, _ = scene_state("VolSDF", 0.01)
    p = pts[0][:M].contiguous()
    out, _ = model.forward_surface(p.to(DEV))
    close(f"bf16x3 sdf M={M}", out, nets.volsdf_forward_surface(sd, p)[0], 1e-4)


def test_point_queries_bf16x3(pts):
    from oracle import nets
    model, _, _ = _model()
    sd, _ = scene_state("VolSDF", 0.01)
    p, v = pts
    rad, sdf, nab = model.forward(p.to(DEV), v.to(DEV))
    r_ref, s_ref, n_ref = nets.volsdf_forward(sd, p, v)
    close("bf16x3 sdf (nabla kernel)", sdf, s_ref, 1e-4)
    close("bf16x3 nabla", nab, n_ref, 1e-3, 1e-3)
    close("bf16x3 radiance", rad, r_ref, 5e-4)
    nm, _, _ = _model("NeuS", None)
    sdn, _ = scene_state("NeuS", None)
    pn = p / 3.0
    rad, sdf, nab = nm.forward(pn.to(DEV), v.to(DEV))
    s_ref, n_ref, f_ref = nets.surface_forward_with_nablas(sdn, pn)
    close("bf16x3 neus sdf", sdf, s_ref, 1e-4)
    close("bf16x3 neus nabla", nab, n_ref, 1e-3, 1e-3)
    close("bf16x3 neus radiance", rad, nets.radiance_forward(sdn, pn, v, n_ref, f_ref, -1, 4), 5e-4)


@pytest.mark.parametrize("sampler", SAMPLERS)
@pytest.mark.parametrize("beta", [0.1, 0.01, 0.002])
def test_fine_sample_bf16x3_vs_reference_golden(golden, beta, sampler):
    """G8 at the precisions the product renders with: Algorithm 1 end to end on the 64 golden rays - iter_usage IDENTICAL to the
    reference's on every ray, beta_map, the 64 fine depths (the assertions tests/test_gpu_parity.py makes of the exact-fp32 path)."""
    from nerfart_amd import hip, rend_util
    model, rk, _ = _model("VolSDF", beta, sampler)
    H, W = int(golden["G9_H"]), int(golden["G9_W"])
    o, d, _ = rend_util.get_rays(tt(golden["G9_c2w"])[None].to(DEV), tt(golden["G9_K"])[None].to(DEV), H, W)
    o, dn = o[0].contiguous(), hip.normalize_dirs(d[0].contiguous())
    blob, prec = model.packed_sampler() or (model.packed()[0], model.precision_id)
    assert prec == (4 if sampler == "fp16x2" else 1)
    alpha, b = model.forward_ab()
    d_fine, beta_map, usage = hip.volsdf_fine_sample(blob, o, dn, 0.0, 6.0, 3.0, float(alpha), float(b), 0.1, 512, 512, 64, 6, 10, precision=prec)
    tag = f"b{beta}"
    same = usage.cpu().numpy() == golden[f"G8_{tag}_iter_usage"]
    print(f"  sampler {sampler or 'bf16x3'}: iter_usage agreement {same.mean():.3f}")
    assert same.all(), "iter_usage identical to the reference's on every golden ray"
    m = torch.from_numpy(same)
    conv = m & (usage.cpu() >= 0)
    bm_ref = tt(golden[f"G8_{tag}_beta_map"])[:, 0]
    close("beta_map (converged rays)", beta_map.cpu()[conv], bm_ref[conv], 1e-7, 1e-5)
    close("beta_map (unconverged rays)", beta_map.cpu()[m & (usage.cpu() < 0)], bm_ref[m & (usage.cpu() < 0)], 0.0, 0.2)
    m = m & ((beta_map.cpu() - bm_ref).abs() <= 1e-4 * bm_ref)
    print(f"  rays compared sample by sample: {int(m.sum())} / {m.numel()}")
    assert m.double().mean() >= 0.95
    close("d_fine", d_fine.cpu()[m], tt(golden[f"G8_{tag}_d_fine"])[m], 3e-4, 0.0, frac=0.99)
    close("d_fine (all)", d_fine.cpu()[m], tt(golden[f"G8_{tag}_d_fine"])[m], 2e-2)


@pytest.mark.parametrize("sampler", SAMPLERS)
@pytest.mark.parametrize("beta,ns", [(0.1, 128), (0.01, 32), (0.01, 128), (0.002, 128)])
def test_volsdf_render_bf16x3_vs_reference_golden(golden, beta, ns, sampler):
    from nerfart_amd import rend_util
    model, rk, render_fn = _model("VolSDF", beta, sampler)
    H, W = int(golden["G9_H"]), int(golden["G9_W"])
    o, d, _ = rend_util.get_rays(tt(golden["G9_c2w"])[None].to(DEV), tt(golden["G9_K"])[None].to(DEV), H, W)
    rgb, depth, ex = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, N_samples=ns, **rk)
    tag = f"G9_b{beta}_n{ns}_"
    same = (ex["iter_usage"][0].cpu().numpy() == golden[tag + "iter_usage"])
    print(f"  rays with identical iter_usage: {same.mean():.3f}")
    assert same.all(), "measured 1.000 on the golden rays at all three beta (round 2)"
    m = torch.from_numpy(same)
    close("rgb (1e-3, every ray with equal rounds)", ex["rgb"][0].cpu()[m], tt(golden[tag + "rgb"])[m], 1e-3)
    # (the 32-spp case has ONE golden ray of 64 whose bisection takes another branch in every arithmetic, fp32 included: test_gpu_parity.py)
    close("rgb (tight, 98%)", ex["rgb"][0].cpu()[m], tt(golden[tag + "rgb"])[m], 2e-4, frac=0.96 if ns == 32 else 0.98)   # measured >= 0.9896
    close("mask", ex["mask_volume"][0].cpu()[m], tt(golden[tag + "mask_volume"])[m], 1e-3)
    close("depth", ex["depth_volume"][0].cpu()[m], tt(golden[tag + "depth_volume"])[m], 1e-2)
    close("normals", ex["normals_volume"][0].cpu()[m], tt(golden[tag + "normals_volume"])[m], 5e-3)
    # rays that took a different number of rounds are still valid renderings of the same field: bounded loosely
    report("rgb (all rays)", ex["rgb"][0].cpu(), tt(golden[tag + "rgb"]))
    close("rgb (all rays, 1e-3)", ex["rgb"][0].cpu(), tt(golden[tag + "rgb"]), 1e-3)                   # north-star bound, every golden ray


@pytest.mark.parametrize("sampler", SAMPLERS)
def test_volsdf_perturb_bf16x3_vs_reference_golden(perturb_golden, sampler):
    """P2 / P3 (perturb=True: the final samples invert the opacity CDF at the reference's recorded uniform numbers) at the product's
    precisions - the assertions tests/test_gpu_parity.py::test_volsdf_perturb_matches_reference_golden makes of the exact-fp32 path."""
    from nerfart_amd import hip, rend_util
    pg = perturb_golden
    model, rk, _ = _model("VolSDF", 0.01, sampler)
    H, W = int(pg["P_H"]), int(pg["P_W"])
    o, d, _ = rend_util.get_rays(tt(pg["P_c2w"])[None].to(DEV), tt(pg["P_K"])[None].to(DEV), H, W)
    o, d = o[0].contiguous(), d[0].contiguous()
    dn = hip.normalize_dirs(d)
    surf_blob, rad_blob = model.packed()
    samp = model.packed_sampler()
    blob, prec = samp or (surf_blob, model.precision_id)
    alpha, b = model.forward_ab()
    d_fine, beta_map, usage = hip.volsdf_fine_sample(blob, o, dn, 0.0, 6.0, 3.0, float(alpha), float(b), 0.1, 512, 512, 64, 6, 10, precision=prec,
                                                     u_final=tt(pg["P2_u_final"]).to(DEV))
    same = usage.cpu().numpy() == pg["P2_iter_usage"]
    print(f"  sampler {sampler or 'bf16x3'}: P2 iter_usage agreement {same.mean():.3f}")
    assert same.mean() >= 0.95
    bm_ref = tt(pg["P2_beta_map"])[:, 0]
    m = torch.from_numpy(same) & ((beta_map.cpu() - bm_ref).abs() <= 1e-4 * bm_ref)
    assert m.double().mean() >= 0.95
    close("d_fine (random u)", d_fine.cpu()[m], tt(pg["P2_d_fine"])[m], 3e-4, 0.0, frac=0.99)
    out = hip.volsdf_render(surf_blob, rad_blob, model.view_tiles, o, d, near=rk["near"], far=rk["far"], R_bg=rk["obj_bounding_radius"],
                            alpha=float(alpha), beta=float(b), max_upsample_steps=rk["max_upsample_steps"], detailed=True, precision=model.precision_id,
                            u_final=tt(pg["P3_u_final"]).to(DEV), sampler=samp)
    same = torch.from_numpy(out["iter_usage"].cpu().numpy() == pg["P3_iter_usage"])
    print(f"  sampler {sampler or 'bf16x3'}: P3 iter_usage agreement {same.double().mean():.3f}")
    assert same.double().mean() >= 0.95
    close("d_vals", out["d_vals"].cpu()[same], tt(pg["P3_d_vals"])[same], 3e-4, 0.0, frac=0.99)
    close("rgb", out["rgb"].cpu()[same], tt(pg["P3_rgb"])[same], 1e-3)
    close("depth", out["depth_volume"].cpu()[same], tt(pg["P3_depth_volume"])[same], 5e-3)


def test_neus_render_bf16x3_vs_reference_golden(golden):
    from nerfart_amd import rend_util
    model, rk, render_fn = _model("NeuS", None)
    H, W = int(golden["G9_H"]), int(golden["G9_W"])
    o, d, _ = rend_util.get_rays(tt(golden["G9_c2w"])[None].to(DEV), tt(golden["G9_K"])[None].to(DEV), H, W)
    rgb, depth, ex = render_fn(o, d, calc_normal=True, detailed_output=True, **rk)
    close("neus rgb", rgb[0], golden["G10_render_rgb"], 1e-3)
    close("neus depth", depth[0], golden["G10_render_depth_volume"], 1e-2)
    close("neus mask", ex["mask_volume"][0], golden["G10_render_mask_volume"], 1e-3)


def test_full_frame_bf16x3_vs_fp32():
    """480 x 270: the two precisions agree pixel for pixel (PSNR) and the fast path keeps the chunk invariance."""
    from nerfart_amd import scene, rend_util
    m32, rk, f32 = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="fp32")
    m16, _, f16 = _model("VolSDF", 0.01)
    H, W = 480, 270
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    kw = {k: v for k, v in rk.items() if k != "rayschunk"}
    a, _, exa = f32(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
    b, _, exb = f16(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
    b2, _, _ = f16(o, d, require_nablas=True, calc_normal=True, detailed_output=False, rayschunk=30000, honor_rayschunk=True, **kw)
    assert torch.equal(b, b2), "chunk invariance"
    same = (exa["iter_usage"] == exb["iter_usage"]).float().mean().item()
    mse = ((a - b) ** 2).mean().item()
    psnr = -10 * np.log10(max(mse, 1e-20))
    err = (a - b).abs()
    print(f"  fp32 vs bf16x3 full frame: identical rounds on {same:.4f} of rays; rgb max {err.max().item():.2e}, "
          f"99.9 pct {err.flatten().kthvalue(int(0.999 * err.numel())).values.item():.2e}, PSNR {psnr:.1f} dB")
    over = int((err.max(dim=-1).values > 1e-3).sum())
    print(f"  rays whose bf16x3 pixel differs from the fp32 pixel by more than 1e-3: {over} of {H * W} ({over / (H * W):.5f})")
    # measured (profiles/r03i_parity_s.log): 0.9966 identical rounds, max 3.5e-3, 87.6 dB; the rays past 1e-3 are rays whose
    # error-bounded sampling took a different branch in the two arithmetics (Algorithm 1 is discontinuous), not arithmetic error
    assert same >= 0.995 and psnr >= 85.0 and err.max().item() < 8e-3
    assert over <= 0.002 * H * W


@pytest.mark.parametrize("M", [1, 127, 128, 1500])
def test_reverse_mode_nabla_bf16x3(pts, M):
    """precision 1 = reverse-mode kernel (k_sdf_grad_bf16); precision 2 = forward-mode tangent quads on the same blob."""
    from oracle import nets
    from nerfart_amd import hip
    model, _, _ = _model()
    sd, _ = scene_state("VolSDF", 0.01)
    surf, _ = model.packed()
    p = pts[0][:M].contiguous()
    pd = p.to(DEV)
    s_ref, n_ref, feat_ref = nets.surface_forward_with_nablas(sd, p)
    d_bg = 3.0 - p.norm(dim=-1)
    s_ref = torch.where(d_bg < s_ref, d_bg, s_ref)
    for rep in range(2):                              # second call reuses the library's scratch (and the L1/L2 state)
        sdf, nab, h7 = hip.sdf_nabla_fwd(surf, pd, 3.0, precision=1)
        close(f"reverse sdf M={M}", sdf, s_ref, 1e-4)
        close(f"reverse nabla M={M}", nab, n_ref, 1e-3, 1e-3)
    sdf2, nab2, h72 = hip.sdf_nabla_fwd(surf, pd, 3.0, precision=2)
    close("reverse vs forward-mode nabla", nab, nab2.cpu(), 5e-4, 5e-4)
    close("reverse vs forward-mode h7", h7, h72.cpu(), 1e-4, 1e-4)
    w8 = nets.folded_weight(sd, "implicit_surface.surface_fc_layers.8")
    b8 = sd["implicit_surface.surface_fc_layers.8.bias"]
    close("reverse h7 -> geometry feature", h7.cpu() @ w8[1:].T + b8[1:], feat_ref, 1e-3, 1e-3)
