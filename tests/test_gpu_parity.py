"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs,
against the golden vectors captured from the real reference, and - at BASELINE.json's full frame size -
through size-independent properties (chunk invariance, sortedness, weight normalisation).

Tolerances (fp32 path, `v_mfma_f32_16x16x4_f32` = exact f32 FMA chains; differences come from summation
order, the hardware exp2/log2 softplus and libm sin/cos):
  * point queries: sdf 2e-5 abs, nabla 2e-4, radiance 1e-4 (measured: ~1e-6, 1e-6, 4e-6);
  * rendered pixels: EVERY ray within 1e-3 (the north_star bound), >= 97% of rays within 1e-4;
  * per-sample arrays (d_vals, sigma, weights ...): >= 99% of entries within the tight tolerance.  VolSDF's
    Algorithm 1 and NeuS' up-sampling are discontinuous in their inputs (a bisection branch flips, an
    inverse-CDF sample sits on a plateau of the CDF), so a 1e-7 difference in one sdf can move a handful of
    samples by 1e-3..1e-2 in depth; the pixels they composite into stay inside the pixel bound.
"""
import os

import numpy as np
import pytest
import torch

from conftest import scene_state, tt

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _oracle():
    from oracle import nets, sampling, render
    return nets, sampling, render


def _model(fw="VolSDF", beta=0.01):
    from nerfart_amd import scene
    model, rk, fn = scene.build_model(fw, seed=0, beta=beta, device=DEV)
    return model, rk, fn


def report(name, a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    err = (a - b).abs()
    print(f"  [{name}] max abs err {err.max().item():.3e}  mean {err.mean().item():.3e}  ref max {b.abs().max().item():.3e}")
    return err


def close(name, a, b, atol, rtol=0.0, frac=1.0):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if a.numel() == 0:
        return
    err = report(name, a, b)
    ok = (err <= atol + rtol * b.abs()) | (a == b)
    if frac < 1.0:
        print(f"    [{name}] fraction within tol: {ok.double().mean().item():.5f} (required {frac})")
    assert ok.double().mean().item() >= frac, f"{name}: {(~ok).sum().item()} / {ok.numel()} outside tol, max {err.max().item():.3e}"


@pytest.fixture(scope="module")
def pts():
    g = torch.Generator().manual_seed(11)
    p = torch.rand(1000, 3, generator=g) * 6 - 3
    p[:300] *= 0.35
    v = torch.nn.functional.normalize(torch.randn(1000, 3, generator=g), dim=-1)
    return p, v


def test_device_is_gfx950():
    assert torch.cuda.is_available()
    name = torch.cuda.get_device_properties(0).gcnArchName
    assert "gfx950" in name, name


@pytest.mark.parametrize("M", [1, 16, 127, 128, 129, 1000])
def test_sdf_fwd_matches_oracle(pts, M):
    nets, _, _ = _oracle()
    model, _, _ = _model()
    sd, _ = scene_state("VolSDF", 0.01)
    p = pts[0][:M].contiguous()
    out, _ = model.forward_surface(p.to(DEV))
    ref = nets.volsdf_forward_surface(sd, p)[0]
    close(f"sdf M={M}", out, ref, 2e-5)
    if M == 1000:
        assert (ref < nets.surface_forward(sd, p)[0]).any(), "inputs must exercise the sphere-background clamp"


def test_sdf_fwd_no_clamp_and_ray_mode(pts):
    nets, _, _ = _oracle()
    from nerfart_amd import hip
    model, _, _ = _model()
    sd, _ = scene_state("VolSDF", 0.01)
    blob, _ = model.packed()
    p = pts[0]
    close("sdf no clamp", hip.sdf_fwd(blob, p.to(DEV), 0.0), nets.surface_forward(sd, p)[0], 2e-5)
    # ray mode with an index list and a padded depth stride
    g = torch.Generator().manual_seed(3)
    R, n, stride = 37, 50, 64
    o = torch.randn(R, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, -2.5])
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    idx = torch.randperm(R, generator=g)[:20].to(torch.int32)
    depth = torch.rand(20, stride, generator=g) * 6
    out = hip.sdf_fwd_rays(blob, o.to(DEV), d.to(DEV), depth.to(DEV), 3.0, ray_idx=idx.to(DEV), n_per_ray=n)
    x = o[idx.long(), None, :] + d[idx.long(), None, :] * depth[:, :n, None]
    ref = nets.volsdf_forward_surface(sd, x.reshape(-1, 3))[0].reshape(20, n)
    close("sdf ray mode", out, ref, 2e-5)


def test_sdf_nabla_and_radiance_match_oracle(pts):
    nets, _, _ = _oracle()
    model, _, _ = _model()
    sd, _ = scene_state("VolSDF", 0.01)
    p, v = pts
    rad, sdf, nab = model.forward(p.to(DEV), v.to(DEV))
    r_ref, s_ref, n_ref = nets.volsdf_forward(sd, p, v)
    close("sdf (nabla kernel)", sdf, s_ref, 2e-5)
    close("nabla", nab, n_ref, 2e-4, 2e-4)
    close("radiance", rad, r_ref, 1e-4)
    # geometry feature reconstructed from h7
    s2, n2, h7 = model.forward_surface_with_nablas(p.to(DEV))
    w8 = nets.folded_weight(sd, "implicit_surface.surface_fc_layers.8")
    feat = h7.cpu() @ w8[1:].T + sd["implicit_surface.surface_fc_layers.8.bias"][1:]
    close("feat from h7", feat, nets.surface_forward(sd, p)[1], 1e-4, 1e-4)


@pytest.mark.parametrize("M", [1, 100, 128, 1000])
def test_fp32_reverse_mode_nabla_matches_the_forward_mode_kernel_and_the_oracle(pts, M):
    """precision 0 = k_sdf_grad (reverse mode, one column per point, transposed chunks of the same blob); precision 3 = the
    forward-mode tangent quads of k_sdf_nabla.  Both multiply exact fp32 products: they agree to summation order."""
    nets, _, _ = _oracle()
    from nerfart_amd import hip
    model, _, _ = _model()
    sd, _ = scene_state("VolSDF", 0.01)
    surf, _ = model.packed()
    p = pts[0][:M].contiguous()
    s_ref, n_ref, _ = nets.surface_forward_with_nablas(sd, p)
    d_bg = 3.0 - p.norm(dim=-1)
    s_ref = torch.where(d_bg < s_ref, d_bg, s_ref)
    for rep in range(2):                               # the second call reuses the library's scratch
        sdf, nab, h7 = hip.sdf_nabla_fwd(surf, p.to(DEV), 3.0, precision=0)
        close(f"reverse sdf M={M}", sdf, s_ref, 2e-5)
        close(f"reverse nabla M={M}", nab, n_ref, 2e-4, 2e-4)
    sdf3, nab3, h73 = hip.sdf_nabla_fwd(surf, p.to(DEV), 3.0, precision=3)
    close("reverse vs forward-mode sdf", sdf, sdf3.cpu(), 1e-6, 1e-6)
    close("reverse vs forward-mode nabla", nab, nab3.cpu(), 2e-5, 2e-5)
    close("reverse vs forward-mode h7", h7, h73.cpu(), 1e-6, 1e-6)
    if M == 1000:                                      # more tiles than workgroups: every workgroup loops, the chunk stream wraps
        g = torch.Generator().manual_seed(5)
        big = (torch.rand(100003, 3, generator=g) * 5 - 2.5).to(DEV)
        a = hip.sdf_nabla_fwd(surf, big, 3.0, precision=0)
        b = hip.sdf_nabla_fwd(surf, big, 3.0, precision=3)
        for name, x, y, tol in (("sdf", a[0], b[0], 1e-6), ("nabla", a[1], b[1], 2e-5), ("h7", a[2], b[2], 1e-6)):
            close(f"100,003 points: reverse vs forward-mode {name}", x, y.cpu(), tol, tol)


def test_neus_point_queries_match_oracle(pts):
    nets, _, _ = _oracle()
    model, _, _ = _model("NeuS", None)
    sd, _ = scene_state("NeuS", None)
    p, v = pts[0] / 3.0, pts[1]
    rad, sdf, nab = model.forward(p.to(DEV), v.to(DEV))
    s_ref, n_ref, f_ref = nets.surface_forward_with_nablas(sd, p)
    close("neus sdf", sdf, s_ref, 2e-5)
    close("neus nabla", nab, n_ref, 2e-4, 2e-4)
    close("neus radiance", rad, nets.radiance_forward(sd, p, v, n_ref, f_ref, -1, 4), 1e-4)
    close("neus sdf only", model.forward_sdf(p.to(DEV)), s_ref, 2e-5)


def test_point_queries_match_reference_golden(golden):
    """straight against vectors captured from the real reference (G3-G5, G10)"""
    model, _, _ = _model()
    p, v = tt(golden["G3_pts"]).to(DEV), tt(golden["G3_view"]).to(DEV)
    close("G5 forward_surface", model.forward_surface(p)[0], golden["G5_forward_surface"], 2e-5)
    rad, sdf, nab = model.forward(p, v)
    close("G5 sdf", sdf, golden["G5_sdf"], 2e-5)
    close("G5 nabla", nab, golden["G5_nabla"], 2e-4, 2e-4)
    close("G5 radiance", rad, golden["G5_radiance"], 1e-4)
    nm, _, _ = _model("NeuS", None)
    rad, sdf, nab = nm.forward(tt(golden["G10_pts"]).to(DEV), tt(golden["G10_view"]).to(DEV))
    close("G10 neus radiance", rad, golden["G10_radiance"], 1e-4)
    close("G10 neus sdf", sdf, golden["G10_sdf"], 2e-5)
    close("G10 neus nabla", nab, golden["G10_nabla"], 2e-4, 2e-4)


def test_get_rays_matches_golden_and_oracle(golden):
    from nerfart_amd import rend_util
    _, _, render = _oracle()
    o, d, inds = rend_util.get_rays(tt(golden["G1_c2w"])[None].to(DEV), tt(golden["G1_K"])[None].to(DEV), 6, 5)
    close("G1 rays_o", o[0], golden["G1_rays_o"], 0)
    close("G1 rays_d", d[0], golden["G1_rays_d"], 2e-6)
    assert torch.equal(inds[0].cpu(), torch.arange(30))
    torch.manual_seed(5)
    o2, d2, sel = rend_util.get_rays(tt(golden["G1_c2w"])[None].to(DEV), tt(golden["G1_K"])[None].to(DEV), 6, 5, N_rays=11)
    ro, rd = render.get_rays(tt(golden["G1_c2w"]), tt(golden["G1_K"]), 6, 5, select_inds=sel[0].cpu())
    close("subset rays_d", d2[0], rd, 2e-6)
    # quaternion pose (the reference's own quaternion branch cannot run - see make_golden.py)
    c2w = tt(golden["G1_c2w"])
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(c2w[:3, :3].double().numpy()).as_quat()      # x, y, z, w
    pose7 = torch.tensor([q[3], q[0], q[1], q[2], *c2w[:3, 3].tolist()], dtype=torch.float32)
    o3, d3, _ = rend_util.get_rays(pose7[None].to(DEV), tt(golden["G1_K"])[None].to(DEV), 6, 5)
    close("quaternion rays_d", d3[0], golden["G1_rays_d"], 2e-5)


@pytest.mark.parametrize("beta", [0.1, 0.01, 0.002])
def test_fine_sample_matches_oracle_and_golden(golden, beta):
    """Algorithm 1 end to end on 64 rays: iter_usage, beta_map and the 64 fine depths."""
    from nerfart_amd import hip, rend_util
    model, rk, _ = _model("VolSDF", beta)
    H, W = int(golden["G9_H"]), int(golden["G9_W"])
    o, d, _ = rend_util.get_rays(tt(golden["G9_c2w"])[None].to(DEV), tt(golden["G9_K"])[None].to(DEV), H, W)
    o, dn = o[0].contiguous(), hip.normalize_dirs(d[0].contiguous())
    blob, _ = model.packed()
    alpha, b = model.forward_ab()
    d_fine, beta_map, usage = hip.volsdf_fine_sample(blob, o, dn, 0.0, 6.0, 3.0, float(alpha), float(b), 0.1, 512, 512, 64, 6, 10)
    tag = f"b{beta}"
    u_ref = golden[f"G8_{tag}_iter_usage"]
    same = usage.cpu().numpy() == u_ref
    print(f"  iter_usage agreement {same.mean():.3f}; hip {np.unique(usage.cpu().numpy(), return_counts=True)} ref {np.unique(u_ref, return_counts=True)}")
    assert same.all(), "fp32 path: iter_usage is identical to the reference's on every golden ray (measured 1.000 in round 1)"
    m = torch.from_numpy(same)
    conv = m & (usage.cpu() >= 0)
    unconv = m & (usage.cpu() < 0)
    close("beta_map (converged rays)", beta_map.cpu()[conv], tt(golden[f"G8_{tag}_beta_map"])[:, 0][conv], 1e-7, 1e-5)
    # never-converged rays carry the bisection's beta+: one flipped comparison moves it by a bisection step
    close("beta_map (unconverged rays)", beta_map.cpu()[unconv], tt(golden[f"G8_{tag}_beta_map"])[:, 0][unconv], 0.0, 0.2)
    # a ray whose bisection took a different branch samples with a different beta+: exclude it below
    bm_ref = tt(golden[f"G8_{tag}_beta_map"])[:, 0]
    m = m & ((beta_map.cpu() - bm_ref).abs() <= 1e-4 * bm_ref)
    print(f"  rays compared sample by sample: {int(m.sum())} / {m.numel()}")
    assert m.double().mean() >= 0.95
    close("d_fine", d_fine.cpu()[m], tt(golden[f"G8_{tag}_d_fine"])[m], 3e-4, 0.0, frac=0.99)
    close("d_fine (all)", d_fine.cpu()[m], tt(golden[f"G8_{tag}_d_fine"])[m], 2e-2)


@pytest.mark.parametrize("beta,ns", [(0.1, 128), (0.01, 32), (0.01, 128), (0.002, 128)])
def test_volsdf_render_matches_reference_golden(golden, beta, ns):
    """render_fn against the reference's own outputs (G9): every extras key."""
    from nerfart_amd import rend_util
    model, rk, render_fn = _model("VolSDF", beta)
    H, W = int(golden["G9_H"]), int(golden["G9_W"])
    o, d, _ = rend_util.get_rays(tt(golden["G9_c2w"])[None].to(DEV), tt(golden["G9_K"])[None].to(DEV), H, W)
    rgb, depth, ex = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, N_samples=ns, **rk)
    tag = f"G9_b{beta}_n{ns}_"
    keys = [k[len(tag):] for k in golden if k.startswith(tag)]
    assert list(ex.keys()) == ["rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_surface", "implicit_nablas",
                               "radiance", "alpha", "p_i", "visibility_weights", "d_vals", "sigma", "beta_map", "iter_usage"]
    same = (ex["iter_usage"][0].cpu().numpy() == golden[tag + "iter_usage"])
    print(f"  rays with identical iter_usage: {same.mean():.3f}")
    assert same.all(), "fp32 path: every golden ray takes the reference's number of up-sampling rounds (measured 1.000 in round 1)"
    m = torch.from_numpy(same)
    tol = {"rgb": (1e-4, 0), "depth_volume": (1e-3, 0), "mask_volume": (1e-4, 0), "normals_volume": (1e-3, 0),
           "implicit_surface": (3e-5, 0), "implicit_nablas": (3e-4, 3e-4), "radiance": (1e-4, 0), "alpha": (2e-4, 0),
           "p_i": (2e-4, 0), "visibility_weights": (2e-4, 0), "d_vals": (3e-4, 0), "sigma": (1e-2, 2e-3),
           "beta_map": (1e-6, 0.2), "iter_usage": (0, 0)}
    # per-sample keys: the floor is the measured fraction of each case (profiles/r03i_parity_s.log) minus a hair - 1.000 on every
    # key at beta 0.1; >= 0.9954 at beta 0.01 n128; >= 0.9906 at beta 0.002; the beta 0.01 n32 case has ONE ray of 64 (1.6 %) whose
    # bisection takes another branch (0.967 .. 0.985 of the entries agree).  Round 2 passed all four at 0.96.
    floor = {(0.1, 128): 1.0, (0.01, 128): 0.995, (0.002, 128): 0.99, (0.01, 32): 0.965}[(beta, ns)]
    for k in keys:
        a, r = tol[k]
        close(k, ex[k][0].cpu()[m], tt(golden[tag + k])[m], a, r, frac=floor)
    # the pixel bound of north_star holds for EVERY ray
    close("rgb (all rays, 1e-3)", ex["rgb"][0].cpu()[m], tt(golden[tag + "rgb"])[m], 1e-3)
    close("mask (all rays, 1e-3)", ex["mask_volume"][0].cpu()[m], tt(golden[tag + "mask_volume"])[m], 1e-3)
    close("depth (all rays, 5e-3)", ex["depth_volume"][0].cpu()[m], tt(golden[tag + "depth_volume"])[m], 5e-3)
    assert rgb.shape == (1, H * W, 3) and depth.shape == (1, H * W)


def test_neus_render_matches_reference_golden(golden):
    from nerfart_amd import rend_util
    model, rk, render_fn = _model("NeuS", None)
    H, W = int(golden["G9_H"]), int(golden["G9_W"])
    o, d, _ = rend_util.get_rays(tt(golden["G9_c2w"])[None].to(DEV), tt(golden["G9_K"])[None].to(DEV), H, W)
    rgb, depth, ex = render_fn(o, d, calc_normal=True, detailed_output=True, **rk)
    tag = "G10_render_"
    tol = {"rgb": 1e-4, "depth_volume": 3e-4, "mask_volume": 1e-4, "normals_volume": 3e-4, "implicit_nablas": 5e-4,
           "implicit_surface": 3e-5, "radiance": 1e-4, "alpha": 3e-4, "cdf": 3e-4, "visibility_weights": 3e-4, "d_final": 3e-4}
    for k in [k[len(tag):] for k in golden if k.startswith(tag)]:
        close(k, ex[k][0], golden[tag + k], tol[k], 3e-4, frac=0.99)
    close("rgb (all rays, 1e-3)", ex["rgb"][0], golden[tag + "rgb"], 1e-3)
    close("depth (all rays, 5e-3)", ex["depth_volume"][0], golden[tag + "depth_volume"], 5e-3)


def test_edge_cases():
    """empty inputs, a single ray, ragged chunking, rays that miss the scene entirely"""
    from nerfart_amd import hip
    model, rk, render_fn = _model()
    blob, _ = model.packed()
    assert hip.sdf_fwd(blob, torch.zeros(0, 3, device=DEV), 3.0).shape == (0,)
    s0, n0, h0 = hip.sdf_nabla_fwd(blob, torch.zeros(0, 3, device=DEV), 3.0)
    assert s0.shape == (0,) and n0.shape == (0, 3) and h0.shape == (0, 256)
    with pytest.raises(RuntimeError, match="precision"):
        hip.sdf_nabla_fwd(blob, torch.zeros(4, 3, device=DEV), 3.0, precision=7)
    # the reverse-mode scratch is the caller's: it comes from PyTorch's caching allocator (visible to its statistics) and the
    # entry point refuses a missing / short one
    import ctypes as C
    torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.max_memory_allocated()
    hip.sdf_nabla_fwd(blob, torch.zeros(256, 3, device=DEV), 3.0, precision=0)
    assert torch.cuda.max_memory_allocated() - before >= hip.lib.nerfart_sdf_nabla_workspace_bytes(0)
    x4 = torch.zeros(4, 3, device=DEV)
    outs = [torch.empty(4, device=DEV), torch.empty(4, 3, device=DEV)]
    small = torch.empty(1024, dtype=torch.uint8, device=DEV)
    rc = hip.lib.nerfart_sdf_nabla_fwd(C.c_void_p(blob.data_ptr()), 0, C.c_void_p(x4.data_ptr()), 4, 3.0, C.c_void_p(outs[0].data_ptr()),
                                       C.c_void_p(outs[1].data_ptr()), None, C.c_void_p(small.data_ptr()), 1024, None)
    assert rc != 0 and "workspace" in hip.lib.nerfart_last_error().decode()
    with pytest.raises(RuntimeError, match="precision"):
        hip.sdf_fwd(blob, torch.zeros(4, 3, device=DEV), 3.0, precision=3)      # the forward-mode cross-check exists for nabla only
    o = torch.tensor([[[0.0, 0.0, -2.5]]], device=DEV)
    d = torch.tensor([[[0.0, 0.0, 1.0]]], device=DEV)
    rgb, depth, ex = render_fn(o, d, require_nablas=True, detailed_output=False, **rk)
    assert rgb.shape == (1, 1, 3) and torch.isfinite(rgb).all() and 0.5 < float(depth) < 3.0
    # a ray pointing away never hits the unit-ish surface: all weight from the sphere background
    rgb2, depth2, ex2 = render_fn(o, -d, require_nablas=True, detailed_output=True, **rk)
    assert torch.isfinite(rgb2).all() and torch.isfinite(ex2["d_vals"]).all()
    # ragged chunking: 100 rays in chunks of 33 == one chunk
    g = torch.Generator().manual_seed(2)
    oo = torch.tensor([0.0, 0.0, -2.5]).expand(100, 3).contiguous().to(DEV)[None]
    dd = (torch.randn(100, 3, generator=g) * 0.25 + torch.tensor([0.0, 0.0, 1.0])).to(DEV)[None]
    a = render_fn(oo, dd, require_nablas=True, detailed_output=False, rayschunk=33, **{k: v for k, v in rk.items() if k != "rayschunk"})
    b = render_fn(oo, dd, require_nablas=True, detailed_output=False, rayschunk=4096, **{k: v for k, v in rk.items() if k != "rayschunk"})
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), "results must not depend on ray chunking"


def test_full_frame_properties():
    """480 x 270 x 128 spp (BASELINE configs[1]): size-independent properties + oracle on a ray subset."""
    from nerfart_amd import scene, rend_util
    nets, sampling, render = _oracle()
    model, rk, render_fn = _model("VolSDF", 0.01)
    H, W = 480, 270
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    kw = {k: v for k, v in rk.items() if k != "rayschunk"}
    rgb, depth, ex = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
    assert rgb.shape == (1, H * W, 3)
    assert torch.isfinite(rgb).all() and rgb.min() >= 0 and rgb.max() <= 1 + 1e-5
    acc = ex["mask_volume"]
    assert acc.min() >= 0 and acc.max() <= 1 + 1e-4
    # chunk invariance at full size (bit exact: every ray's arithmetic is independent of its neighbours)
    rgb2, depth2, _ = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=False, rayschunk=40000, honor_rayschunk=True, **kw)
    assert torch.equal(rgb, rgb2) and torch.equal(depth, depth2)
    # detailed pass on a strided subset: sorted depths, weights sum to acc, oracle agreement
    sel = torch.arange(0, H * W, 2025)[:64]
    ro, rd = o[:, sel], d[:, sel]
    rgb_s, depth_s, ex_s = render_fn(ro, rd, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
    assert torch.equal(rgb_s, rgb[:, sel]), "a ray renders identically alone and inside the full frame"
    dv = ex_s["d_vals"][0]
    assert (dv[:, 1:] >= dv[:, :-1]).all()
    close("sum tau == acc", ex_s["visibility_weights"][0].sum(-1), ex_s["mask_volume"][0], 1e-5)
    sd, _ = scene_state("VolSDF", 0.01)
    with torch.no_grad():
        ref = render.volsdf_render(sd, ro[0].cpu(), rd[0].cpu(), near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=128, max_upsample_steps=6)
    same = (ex_s["iter_usage"][0].cpu() == ref["iter_usage"])
    print("  full-frame subset: identical iter_usage on", same.double().mean().item())
    assert same.double().mean().item() >= 0.98, "fp32 path: measured 1.000 (64 of 64 rays take the CPU's number of rounds)"
    close("rgb vs oracle", rgb_s[0].cpu()[same], ref["rgb"][same], 1e-4, frac=0.97)
    close("rgb vs oracle (all rays, 1e-3)", rgb_s[0].cpu()[same], ref["rgb"][same], 1e-3)
    close("depth vs oracle", depth_s[0].cpu()[same], ref["depth_volume"][same], 5e-3)
    u = ex_s["iter_usage"][0].cpu()
    print("  iter_usage histogram (subset):", torch.unique(u, return_counts=True))


def test_implicit_surface_forward_as_the_reference_consumers_call_it(pts):
    """mesh_util.extract_mesh (:110) and ray_casting.sphere_tracing (:179) call model.implicit_surface.forward(pts) /
    forward(pts, return_h=True) / forward_with_nablas(pts): same values as the oracle, no sphere clamp."""
    from oracle import nets
    from nerfart_amd import scene
    sd, _ = scene_state("VolSDF", 0.01)
    model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV)
    p = pts[0][:257].contiguous()
    s_ref, f_ref = nets.surface_forward(sd, p)
    with torch.no_grad():
        close("implicit_surface.forward", model.implicit_surface.forward(p.to(DEV)), s_ref, 2e-6)
        s2, f2 = model.implicit_surface.forward(p.to(DEV), return_h=True)
        close("forward(return_h) sdf", s2, s_ref, 2e-6)
        close("forward(return_h) feature", f2, f_ref, 2e-5)
        s3, n3, f3 = model.implicit_surface.forward_with_nablas(p.reshape(1, 257, 3).to(DEV))
        _, n_ref, _ = nets.surface_forward_with_nablas(sd, p)
        assert s3.shape == (1, 257) and n3.shape == (1, 257, 3) and f3.shape == (1, 257, 256)
        close("forward_with_nablas nablas", n3[0], n_ref, 5e-6)


# ---- perturb=True: the samplers invert their CDFs at the caller's uniform numbers -----------------------------------
def test_volsdf_perturb_matches_reference_golden(perturb_golden):
    """fine_sample / volume_render with perturb=True against the reference's outputs for the uniform numbers it drew
    (tests/golden/make_golden_perturb.py)."""
    from nerfart_amd import hip, rend_util
    pg = perturb_golden
    model, rk, _ = _model("VolSDF", 0.01)
    H, W = int(pg["P_H"]), int(pg["P_W"])
    o, d, _ = rend_util.get_rays(tt(pg["P_c2w"])[None].to(DEV), tt(pg["P_K"])[None].to(DEV), H, W)
    o, d = o[0].contiguous(), d[0].contiguous()
    dn = hip.normalize_dirs(d)
    surf_blob, rad_blob = model.packed()
    alpha, b = model.forward_ab()
    d_fine, beta_map, usage = hip.volsdf_fine_sample(surf_blob, o, dn, 0.0, 6.0, 3.0, float(alpha), float(b), 0.1, 512, 512, 64, 6, 10,
                                                     u_final=tt(pg["P2_u_final"]).to(DEV))
    same = usage.cpu().numpy() == pg["P2_iter_usage"]
    assert same.mean() >= 0.95
    bm_ref = tt(pg["P2_beta_map"])[:, 0]
    m = torch.from_numpy(same) & ((beta_map.cpu() - bm_ref).abs() <= 1e-4 * bm_ref)
    assert m.double().mean() >= 0.95
    close("d_fine (random u)", d_fine.cpu()[m], tt(pg["P2_d_fine"])[m], 3e-4, 0.0, frac=0.99)
    out = hip.volsdf_render(surf_blob, rad_blob, model.view_tiles, o, d, near=rk["near"], far=rk["far"], R_bg=rk["obj_bounding_radius"],
                            alpha=float(alpha), beta=float(b), max_upsample_steps=rk["max_upsample_steps"], detailed=True,
                            u_final=tt(pg["P3_u_final"]).to(DEV))
    same = torch.from_numpy(out["iter_usage"].cpu().numpy() == pg["P3_iter_usage"])
    assert same.double().mean() >= 0.95
    close("d_vals", out["d_vals"].cpu()[same], tt(pg["P3_d_vals"])[same], 3e-4, 0.0, frac=0.99)
    close("rgb", out["rgb"].cpu()[same], tt(pg["P3_rgb"])[same], 1e-3)
    close("depth", out["depth_volume"].cpu()[same], tt(pg["P3_depth_volume"])[same], 5e-3)


def test_neus_perturb_matches_reference_golden(perturb_golden):
    from nerfart_amd import hip, rend_util
    pg = perturb_golden
    model, rk, _ = _model("NeuS", None)
    H, W = int(pg["P_H"]), int(pg["P_W"])
    o, d, _ = rend_util.get_rays(tt(pg["P_c2w"])[None].to(DEV), tt(pg["P_K"])[None].to(DEV), H, W)
    surf_blob, rad_blob = model.packed()
    out = hip.neus_render(surf_blob, rad_blob, model.view_tiles, o[0].contiguous(), d[0].contiguous(),
                          obj_bounding_radius=rk["obj_bounding_radius"], s=float(model.forward_s()), n_upsample_iters=rk["N_upsample_iters"],
                          calc_normal=True, detailed=True, u_new=tt(pg["P4_u_new"]).to(DEV))
    close("d_final", out["d_final"], pg["P4_d_final"], 3e-4, 3e-4, frac=0.99)
    close("implicit_surface", out["implicit_surface"], pg["P4_implicit_surface"], 3e-5, 3e-4, frac=0.99)
    close("rgb", out["rgb"], pg["P4_rgb"], 1e-3)
    close("depth", out["depth_volume"], pg["P4_depth_volume"], 5e-3)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("algo", ["direct_use", "direct_more"])
def test_neus_direct_upsampling_matches_reference_golden(neus_algos_golden, algo, precision):
    """G10b: NeuS `upsample_algo` = 'direct_use' / 'direct_more' (neus.py:242-269, YAML-reachable through model.upsample_algo :735) through
    render_fn against the reference's outputs, at G10's tolerances - every extras key at perturb=False; rgb / depth / d_final at
    perturb=True with the reference's recorded uniform numbers (`uniforms=`); calc_normal=False takes the sampler's own sdf row."""
    from nerfart_amd import scene, rend_util
    ag = neus_algos_golden
    model, rk, render_fn = scene.build_model("NeuS", seed=0, beta=None, device=DEV, precision=precision)
    H, W = int(ag["A_H"]), int(ag["A_W"])
    o, d, _ = rend_util.get_rays(tt(ag["A_c2w"])[None].to(DEV), tt(ag["A_K"])[None].to(DEV), H, W)
    kw = dict(rk, upsample_algo=algo)
    rgb, depth, ex = render_fn(o, d, calc_normal=True, detailed_output=True, **kw)
    tag = f"A_{algo}_"
    loose = 1.0 if precision == "fp32" else 3.0           # split-bf16: ~2^-16 relative per product instead of fp32 round-off
    tol = {"rgb": 1e-4, "depth_volume": 3e-4, "mask_volume": 1e-4, "normals_volume": 3e-4, "implicit_nablas": 5e-4,
           "implicit_surface": 3e-5, "radiance": 1e-4, "alpha": 3e-4, "cdf": 3e-4, "visibility_weights": 3e-4, "d_final": 3e-4}
    assert [k for k in ex.keys() if k != "d_all"] == ["rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_nablas", "implicit_surface",
                                                       "radiance", "alpha", "cdf", "visibility_weights", "d_final"]
    for k, t in tol.items():
        close(f"{algo} {k}", ex[k][0], ag[tag + k], loose * t, 3e-4, frac=0.99)
    close("rgb (all rays, 1e-3)", ex["rgb"][0], ag[tag + "rgb"], 1e-3)
    close("depth (all rays, 5e-3)", ex["depth_volume"][0], ag[tag + "depth_volume"], 5e-3)
    rgb2, depth2, ex2 = render_fn(o, d, calc_normal=False, detailed_output=False, **kw)
    assert torch.equal(rgb2, rgb) and torch.equal(depth2, depth), "pixels must not depend on calc_normal"
    rgb_p, depth_p, ex_p = render_fn(o, d, calc_normal=True, detailed_output=True, perturb=True, uniforms=tt(ag[tag + "perturb_u"]).to(DEV),
                                     **{k: v for k, v in kw.items() if k != "perturb"})
    close("perturb d_final", ex_p["d_final"][0], ag[tag + "perturb_d_final"], 3e-4, 3e-4, frac=0.99)
    close("perturb rgb", rgb_p[0], ag[tag + "perturb_rgb"], 1e-3)
    close("perturb depth", depth_p[0], ag[tag + "perturb_depth_volume"], 5e-3)
    close("perturb mask", ex_p["mask_volume"][0], ag[tag + "perturb_mask_volume"], 1e-3)


@pytest.mark.parametrize("algo", ["official_solution", "direct_use", "direct_more"])
def test_neus_sampler_on_its_own_equals_the_renderers_depths(algo):
    """hip.neus_sample (the NeuS sampler on the stage entry points: what pass 2 of a perturb=True NeuS fine-tune step calls) returns the fused
    renderer's `d_all` bit for bit - deterministic and with given uniform numbers, fp32 and split-bf16, a ray count off every tile size."""
    from nerfart_amd import hip, scene, rend_util
    H, W = 19, 13
    c2w, K = scene.camera(H, W)
    g = torch.Generator().manual_seed(8)
    u = torch.rand(H * W, 64, generator=g).to(DEV)
    for precision in ("fp32", "bf16x3"):
        model, rk, _ = scene.build_model("NeuS", seed=0, beta=None, device=DEV, precision=precision)
        o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
        o, d = o[0].contiguous(), d[0].contiguous()
        surf, rad = model.packed()
        for un in (None, u):
            ref = hip.neus_render(surf, rad, model.view_tiles, o, d, obj_bounding_radius=rk["obj_bounding_radius"], s=float(model.forward_s()),
                                  n_upsample_iters=rk["N_upsample_iters"], calc_normal=False, detailed=True, precision=model.precision_id, u_new=un,
                                  upsample_algo=algo)["d_all"]
            got = hip.neus_sample(surf, o, d, obj_bounding_radius=rk["obj_bounding_radius"], n_upsample_iters=rk["N_upsample_iters"],
                                  precision=model.precision_id, u_new=un, upsample_algo=algo)
            assert got.shape == ref.shape == (H * W, 128) and torch.equal(got, ref), (algo, precision, un is not None)
    assert hip.neus_sample(surf, o[:0], d[:0], obj_bounding_radius=1.0).shape == (0, 128)


def test_render_fn_perturb_draws_fresh_samples():
    """render_fn(perturb=True) (the reference's training default, volsdf.py:982): two calls draw different final samples,
    the coarse samples stay, and the image stays close to the deterministic render (it is the same quadrature rule)."""
    from nerfart_amd import rend_util, scene
    for fw, key in (("VolSDF", "d_vals"), ("NeuS", "d_all")):
        model, rk, render_fn = _model(fw, 0.01 if fw == "VolSDF" else None)
        c2w, K = scene.camera(8, 8)
        o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), 8, 8)
        kw = dict(rk); kw["perturb"] = False
        _, _, ex0 = render_fn(o, d, detailed_output=True, **kw)
        kw["perturb"] = True
        torch.manual_seed(1)
        _, _, ex1 = render_fn(o, d, detailed_output=True, **kw)
        _, _, ex2 = render_fn(o, d, detailed_output=True, **kw)
        assert not torch.equal(ex1[key], ex2[key]) and not torch.equal(ex1[key], ex0[key])
        assert torch.all(ex1[key][..., 1:] >= ex1[key][..., :-1])
        assert float((ex1["rgb"] - ex0["rgb"]).abs().max()) < 0.1


def test_dataset_to_frames(tmp_path):
    """The data-side rows feeding the renderer end to end (render.py:286-330, :527-533): scene folder -> SceneDataset ->
    spiral camera path -> get_rays -> render_fn -> H x W x 3 frames."""
    from PIL import Image
    from nerfart_amd import camera_path, dataio, rend_util
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "campath_golden.npz"))
    os.makedirs(tmp_path / "images"); os.makedirs(tmp_path / "matte")
    cams = {}
    for i in range(4):
        Image.fromarray(np.full((96, 54, 3), 40 * i, np.uint8)).save(tmp_path / "images" / f"{i:06d}.png")
        Image.fromarray(np.full((96, 54, 3), 255, np.uint8)).save(tmp_path / "matte" / f"{i:06d}.png")
        cams[f"world_mat_{i}"], cams[f"scale_mat_{i}"] = z[f"C2_world_mat_{i}"], z[f"C2_scale_mat_{i}"]
    np.savez(tmp_path / "cameras.npz", **cams)
    ds = dataio.SceneDataset(False, str(tmp_path), downscale=2, scale_radius=3.0)
    assert (ds.H, ds.W) == (48, 27)
    K, H, W = camera_path.render_intrinsics(ds, H=24, W=16)
    c2ws = torch.stack(ds.c2w_all, 0).numpy()
    path = camera_path.spiral_path(c2ws, 3, rot_percentile=85, rot_rad=0.3)
    model, rk, render_fn = _model("VolSDF", 0.01)
    for c2w in path:
        o, d, _ = rend_util.get_rays(torch.from_numpy(c2w).float().to(DEV)[None], K.to(DEV)[None], H, W)
        rgb, depth, ex = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=False, **rk)
        img = (rgb.cpu().reshape(H, W, 3).numpy() * 255.0).astype(np.uint8)
        assert img.shape == (24, 16, 3) and torch.isfinite(rgb).all() and torch.isfinite(depth).all()
        assert ex["normals_volume"].shape[-2:] == (H * W, 3)


def test_batchify_query_over_the_hip_point_query(pts):
    """Row a10: `batchify_query(model.forward, pts[(B),R,P,3], view_dirs, chunk=netchunk, dim_batchify=1, return_nablas=True)` -
    the reference's call at volsdf.py:506-513 - on the HIP-backed model equals the oracle's point query, for any chunk."""
    from oracle import nets
    from nerfart_amd.train_util import batchify_query
    model, _, _ = _model()
    sd, _ = scene_state("VolSDF", 0.01)
    p, v = pts
    x, vd = p[:960].reshape(1, 40, 24, 3).to(DEV), v[:960].reshape(1, 40, 24, 3).to(DEV)
    rad, sdf, nab = batchify_query(model.forward, x, vd, chunk=333, dim_batchify=1, return_nablas=True)
    assert rad.shape == (1, 40, 24, 3) and sdf.shape == (1, 40, 24) and nab.shape == (1, 40, 24, 3)
    r_ref, s_ref, n_ref = nets.volsdf_forward(sd, p[:960], v[:960])
    close("batchify sdf", sdf.reshape(-1), s_ref, 2e-5)
    close("batchify nabla", nab.reshape(-1, 3), n_ref, 2e-4, 2e-4)
    close("batchify radiance", rad.reshape(-1, 3), r_ref, 1e-4)
    rad1, sdf1, nab1 = batchify_query(model.forward, x, vd, chunk=1 << 20, dim_batchify=1, return_nablas=True)
    assert torch.equal(rad, rad1) and torch.equal(sdf, sdf1) and torch.equal(nab, nab1), "results do not depend on netchunk"
    sdf_only = batchify_query(lambda q, return_nablas: model.forward_surface(q)[0], x, chunk=500, dim_batchify=1, return_nablas=False)
    close("batchify forward_surface", sdf_only.reshape(-1), s_ref, 2e-5)


def test_nabla_entry_point_is_reentrant_across_streams():
    """kernels must be re-entrant per device and honour the current stream.  The reverse-mode scratch is the caller's
    (one workspace per call from the caching allocator), so two streams may run nerfart_sdf_nabla_fwd at the same time: results
    equal the sequential ones bit for bit, in both precisions."""
    from nerfart_amd import hip, scene
    g = torch.Generator().manual_seed(11)
    xs = [(torch.rand(200_000, 3, generator=g) * 4 - 2).to(DEV) for _ in range(2)]
    for precision, name in ((0, "fp32"), (1, "bf16x3")):
        model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision=name)
        blob, _ = model.packed()
        ref = [hip.sdf_nabla_fwd(blob, x, 3.0, precision=precision) for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        out = [None, None]
        for rep in range(3):
            for i, st in enumerate(streams):
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    out[i] = hip.sdf_nabla_fwd(blob, xs[i], 3.0, precision=precision)
            torch.cuda.synchronize()
            for i in range(2):
                for a, b in zip(out[i], ref[i]):
                    assert torch.equal(a, b), f"{name}, stream {i}, repetition {rep}: concurrent call differs from the sequential one"


def test_renderer_is_reentrant_across_threads_and_streams():
    """The fused renderer from two host threads, each on its own stream (ctypes releases the GIL inside the entry point; every
    up-sampling round has a host read): the Python host hands each stream its own workspace, so the two halves of a frame rendered
    concurrently equal the sequential render bit for bit.  (A workspace shared per DEVICE raced: tools/archive/exp_two_stream.py.)"""
    import threading
    from nerfart_amd import scene, rend_util
    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
    H, W = 96, 54
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    kw = {k: v for k, v in rk.items() if k != "rayschunk"}
    args = dict(require_nablas=True, calc_normal=True, detailed_output=False, **kw)
    ref_rgb, ref_depth, _ = render_fn(o, d, **args)
    torch.cuda.synchronize()
    N = o.shape[1]
    bounds = [(0, N // 2), (N // 2, N)]
    for rep in range(3):
        out, err = [None, None], []
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        ev = torch.cuda.Event()
        ev.record()

        def work(t):
            try:
                with torch.cuda.stream(streams[t]):
                    streams[t].wait_event(ev)
                    a, b = bounds[t]
                    out[t] = render_fn(o[:, a:b].contiguous(), d[:, a:b].contiguous(), **args)[:2]
            except Exception as e:                                  # surface failures of the worker threads
                err.append(e)
        th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        assert not err, err
        rgb = torch.cat([out[0][0], out[1][0]], dim=1)
        depth = torch.cat([out[0][1], out[1][1]], dim=1)
        assert torch.equal(rgb, ref_rgb) and torch.equal(depth, ref_depth), f"repetition {rep}"
