"""Surface renderer + SDF volume on the HIP path against the reference's goldens (raycast_golden.npz) and the
oracle.  The masks are decisions on the sign of an SDF value: a ray whose marched value sits within fp32 round-off of zero may
flip; the goldens were chosen without such rays (asserted: masks identical)."""
import os

import numpy as np
import pytest
import torch

from conftest import scene_state, tt

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raycast_golden.npz")


@pytest.fixture(scope="module")
def rg():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


def _close(name, a, b, atol):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), name
    err = (a[fin] - b[fin]).abs().max().item() if fin.any() else 0.0
    print(f"  [{name}] max abs err {err:.3e}")
    assert err <= atol, (name, err)


@pytest.mark.parametrize("fw,precision", [("VolSDF", "fp32"), ("VolSDF", "bf16x3"), ("NeuS", "fp32"), ("NeuS", "bf16x3")])
def test_ray_casting_matches_reference_goldens(rg, fw, precision):
    from nerfart_amd import scene, ray_casting as rc
    model, _, _ = scene.build_model(fw, seed=0, beta=0.01 if fw == "VolSDF" else None, device=DEV, precision=precision)
    o, d = tt(rg[f"{fw}_rays_o"]).to(DEV)[None], tt(rg[f"{fw}_rays_d"]).to(DEV)[None]
    dn = torch.nn.functional.normalize(d, dim=-1)
    near, far = float(rg[f"{fw}_near"]), float(rg[f"{fw}_far"])
    tol = 2e-5 if precision == "fp32" else 2e-4
    for tau in (0.0, 0.02):
        k = f"{fw}_root_tau{tau}_"
        depth, pts, mask, msc = rc.root_finding_surface_points(model.implicit_surface, o, dn, near=near, far=far, N_steps=256, logit_tau=tau,
                                                               N_secant_steps=8, fill_inf=(tau == 0.0))
        assert depth.shape == (1, 90) and pts.shape == (1, 90, 3) and mask.dtype == torch.bool
        assert np.array_equal(mask[0].cpu().numpy(), rg[k + "mask"]) and np.array_equal(msc[0].cpu().numpy(), rg[k + "mask_sign_change"])
        _close(k + "d", depth[0], rg[k + "d"], tol)
        _close(k + "pt", pts[0], rg[k + "pt"], tol)
    depth, pts, live = rc.sphere_tracing_surface_points(model.implicit_surface, o, dn, near=near, far=far, N_iters=20)
    assert np.array_equal(live[0].cpu().numpy(), rg[f"{fw}_sphere_mask"])
    _close("sphere d", depth[0], rg[f"{fw}_sphere_d"], 10 * tol)
    _close("sphere pt", pts[0], rg[f"{fw}_sphere_pt"], 10 * tol)
    # per-ray near / far tensors and the unbatched call shape give the same result
    d2, p2, m2, _ = rc.root_finding_surface_points(model.implicit_surface, o[0], dn[0], near=torch.full((90,), near, device=DEV),
                                                   far=torch.full((90,), far, device=DEV), batched=False)
    d1, p1, m1, _ = rc.root_finding_surface_points(model.implicit_surface, o, dn, near=near, far=far)
    assert d2.shape == (90,) and torch.equal(d2, d1[0]) and torch.equal(p2, p1[0]) and torch.equal(m2, m1[0])
    for algo, cfgs in (("root_finding", dict(near=near, far=far, N_steps=256, N_secant_steps=8)), ("sphere_tracing", dict(near=near, far=far, N_iters=20))):
        col, dep, ex = rc.surface_render(o, d, model, calc_normal=True, rayschunk=37, ray_casting_algo=algo, ray_casting_cfgs=cfgs)
        k = f"{fw}_render_{algo}_"
        assert list(ex.keys()) == ["implicit_nablas", "mask_surface", "normals_surface"]
        col1, dep1, ex1 = rc.surface_render(o, d, model, calc_normal=True, rayschunk=4096, ray_casting_algo=algo, ray_casting_cfgs=cfgs)
        assert torch.equal(col, col1) and torch.equal(dep, dep1) and torch.equal(ex["mask_surface"], ex1["mask_surface"]), \
            "rays are marched in slices of rayschunk: 37-ray slices == one slice"
        m = ex["mask_surface"][0].cpu()
        assert np.array_equal(m.numpy(), rg[k + "mask_surface"])
        _close(k + "rgb", col[0], rg[k + "rgb"], 1e-4 if precision == "fp32" else 1e-3)
        _close(k + "depth", dep[0], rg[k + "depth"], 10 * tol)
        _close(k + "normals", ex["normals_surface"][0], rg[k + "normals_surface"], 1e-3 if precision == "fp32" else 5e-3)
        _close(k + "nablas (hit rays)", ex["implicit_nablas"][0].cpu()[m], rg[k + "implicit_nablas"][m.numpy()], 1e-3 if precision == "fp32" else 5e-3)
    with pytest.raises(TypeError):
        rc.root_finding_surface_points(lambda x: x[..., 0], o, dn)


def test_full_frame_surface_render_and_sdf_volume():
    """480 x 270 surface render (render.py --use_surface_render): properties + oracle on a subset; 64^3 SDF volume vs the oracle."""
    from nerfart_amd import scene, rend_util, ray_casting as rc, mesh_util
    from oracle import raycast, nets
    model, rk, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
    H, W = 480, 270
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(DEV), K[None].to(DEV), H, W)
    cfgs = dict(near=0.0, far=6.0, N_steps=256, N_secant_steps=8)
    col, dep, ex = rc.surface_render(o, d, model, ray_casting_algo="root_finding", ray_casting_cfgs=cfgs)
    m = ex["mask_surface"]
    assert col.shape == (1, H * W, 3) and m.float().mean() > 0.3 and torch.isfinite(col).all()
    assert (col[~m] == 0).all() and torch.isinf(dep[~m]).all() and (dep[m] > 0.5).all() and (dep[m] < 4.0).all()
    # the hit points lie on the zero level set
    sdf_hit = model.implicit_surface.forward((o + torch.nn.functional.normalize(d, dim=-1) * dep[..., None])[m])
    print(f"  full frame: {int(m.sum())} hits; |sdf| at the hit points max {sdf_hit.abs().max().item():.2e}")
    assert sdf_hit.abs().max() < 1e-4
    sel = torch.arange(0, H * W, 2025)[:64]
    sd, _ = scene_state("VolSDF", 0.01)
    ref = raycast.surface_render(sd, o[0, sel].cpu(), d[0, sel].cpu(), "root_finding", near=0.0, far=6.0, n_steps=256, n_secant=8)
    same = ref["mask_surface"] == m[0, sel].cpu()
    assert same.float().mean() >= 0.98
    hit = same & ref["mask_surface"]
    _close("depth vs oracle", dep[0, sel].cpu()[hit], ref["depth"][hit], 2e-4)
    _close("rgb vs oracle", col[0, sel].cpu()[hit], ref["rgb"][hit], 1e-3)
    # SDF volume (mesh_util.extract_mesh's grid sweep)
    N = 64
    vol = mesh_util.sdf_volume(model.implicit_surface, volume_size=2.0, N=N, chunk=100000)
    assert vol.shape == (N, N, N)
    pts = mesh_util.grid_points(N, 2.0, "cpu")
    ref_v = nets.surface_forward(sd, pts)[0].reshape(N, N, N)
    _close("sdf volume 64^3", vol, ref_v, 1e-4)
    assert (vol[N // 2, N // 2, N // 2] < 0) and (vol[0, 0, 0] > 0)                       # inside at the centre, outside at the corner


@pytest.mark.parametrize("algo", ["root_finding", "sphere_tracing"])
def test_surface_render_accepts_an_empty_ray_batch(algo):
    """A rank (or a mask selection) with no rays: zero-sized outputs, as the kernels' n <= 0 path gives everywhere else."""
    from nerfart_amd import scene, ray_casting as rc
    model, _, _ = scene.build_model("VolSDF", seed=0, beta=0.01, device=DEV, precision="bf16x3")
    o = torch.zeros(1, 0, 3, device=DEV)
    cfgs = dict(near=0.0, far=6.0, N_steps=64, N_secant_steps=4) if algo == "root_finding" else dict(near=0.0, far=6.0, N_iters=10)
    col, dep, ex = rc.surface_render(o, o.clone(), model, calc_normal=True, ray_casting_algo=algo, ray_casting_cfgs=cfgs)
    assert col.shape == (1, 0, 3) and dep.shape == (1, 0) and ex["mask_surface"].shape == (1, 0) and ex["normals_surface"].shape == (1, 0, 3)
