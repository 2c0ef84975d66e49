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


def _model(fw="VolSDF", beta=0.01):
    from nerfart_amd import scene
    return scene.build_model(fw, seed=0, beta=beta, device=DEV, precision="bf16x3")


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


@pytest.mark.parametrize("beta,ns", [(0.1, 128), (0.01, 128), (0.002, 128)])
def test_volsdf_render_bf16x3_vs_reference_golden(golden, beta, ns):
    from nerfart_amd import rend_util
    model, rk, render_fn = _model("VolSDF", beta)
    H, W = int(golden["G9_H"]), int(golden["G9_W"])
    o, d, _ = rend_util.get_rays(tt(golden["G9_c2w"])[None].to(DEV), tt(golden["G9_K"])[None].to(DEV), H, W)
    rgb, depth, ex = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, N_samples=ns, **rk)
    tag = f"G9_b{beta}_n{ns}_"
    same = (ex["iter_usage"][0].cpu().numpy() == golden[tag + "iter_usage"])
    print(f"  rays with identical iter_usage: {same.mean():.3f}")
    assert same.all(), "measured 1.000 on the golden rays at all three beta (round 2)"
    m = torch.from_numpy(same)
    close("rgb (1e-3, every ray with equal rounds)", ex["rgb"][0].cpu()[m], tt(golden[tag + "rgb"])[m], 1e-3)
    close("rgb (tight, 98%)", ex["rgb"][0].cpu()[m], tt(golden[tag + "rgb"])[m], 2e-4, frac=0.98)       # measured >= 0.9896
    close("mask", ex["mask_volume"][0].cpu()[m], tt(golden[tag + "mask_volume"])[m], 1e-3)
    close("depth", ex["depth_volume"][0].cpu()[m], tt(golden[tag + "depth_volume"])[m], 1e-2)
    close("normals", ex["normals_volume"][0].cpu()[m], tt(golden[tag + "normals_volume"])[m], 5e-3)
    # rays that took a different number of rounds are still valid renderings of the same field: bounded loosely
    report("rgb (all rays)", ex["rgb"][0].cpu(), tt(golden[tag + "rgb"]))
    close("rgb (all rays, 1e-3)", ex["rgb"][0].cpu(), tt(golden[tag + "rgb"]), 1e-3)                   # north-star bound, every golden ray


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
    b2, _, _ = f16(o, d, require_nablas=True, calc_normal=True, detailed_output=False, rayschunk=30000, **kw)
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
