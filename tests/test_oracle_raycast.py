"""The ray-casting oracle (oracle/raycast.py) against the reference's own outputs (tests/golden/make_golden_raycast.py), CPU."""
import os

import numpy as np
import pytest
import torch

from conftest import scene_state, tt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raycast_golden.npz")


@pytest.fixture(scope="module")
def rg():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("fw", ["VolSDF", "NeuS"])
def test_root_finding_sphere_tracing_and_surface_render_match_reference(rg, fw):
    from oracle import raycast
    sd, _ = scene_state(fw, 0.01 if fw == "VolSDF" else None)
    o, d = tt(rg[f"{fw}_rays_o"]), tt(rg[f"{fw}_rays_d"])
    dn = torch.nn.functional.normalize(d, dim=-1)
    near, far = float(rg[f"{fw}_near"]), float(rg[f"{fw}_far"])
    for tau in (0.0, 0.02):
        k = f"{fw}_root_tau{tau}_"
        depth, pts, mask, msc = raycast.root_finding(sd, o, dn, near, far, 256, tau, 8, fill_inf=(tau == 0.0))
        assert np.array_equal(mask.numpy(), rg[k + "mask"]) and np.array_equal(msc.numpy(), rg[k + "mask_sign_change"])
        np.testing.assert_allclose(depth.numpy(), rg[k + "d"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(pts.numpy(), rg[k + "pt"], rtol=0, atol=2e-6)
        assert mask.sum() >= 40 and not bool(mask[1])                                       # ray 1 points away from the object
        if fw == "VolSDF":
            assert float(depth[0]) == 0.0                                                   # ray 0 starts inside the surface
    depth, pts, live = raycast.sphere_tracing(sd, o, dn, near, far, 20)
    assert np.array_equal(live.numpy(), rg[f"{fw}_sphere_mask"])
    np.testing.assert_allclose(depth.numpy(), rg[f"{fw}_sphere_d"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(pts.numpy(), rg[f"{fw}_sphere_pt"], rtol=0, atol=5e-6)
    for algo, cfgs in (("root_finding", dict(near=near, far=far, n_steps=256, n_secant=8)), ("sphere_tracing", dict(near=near, far=far, n_iters=20))):
        out = raycast.surface_render(sd, o, d, algo, rad_multires_view=-1 if fw == "VolSDF" else 4, **cfgs)
        k = f"{fw}_render_{algo}_"
        assert np.array_equal(out["mask_surface"].numpy(), rg[k + "mask_surface"])
        m = out["mask_surface"]
        np.testing.assert_allclose(out["rgb"].numpy(), rg[k + "rgb"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(out["depth"].numpy(), rg[k + "depth"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(out["normals_surface"].numpy(), rg[k + "normals_surface"], rtol=0, atol=2e-5)
        # nablas at unhit rays are evaluated at the fill point (1, 1, 1): compared everywhere
        np.testing.assert_allclose(out["implicit_nablas"].numpy()[m], rg[k + "implicit_nablas"][m.numpy()], rtol=0, atol=2e-5)


def test_grid_points_regular_and_reference_shear():
    from nerfart_amd import mesh_util
    N, s = 5, 2.0
    p = mesh_util.grid_points(N, s, "cpu").numpy()
    assert p.shape == (125, 3)
    ax = np.linspace(-1.0, 1.0, N, dtype=np.float32)
    ref = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)            # x slowest, z fastest
    np.testing.assert_allclose(p, ref, atol=1e-7)
    # the reference's arithmetic (mesh_util.py:87-102), restated with float64 as numpy does
    idx = np.arange(N ** 3)
    xyz = np.zeros([N ** 3, 3])
    xyz[:, 2] = idx % N; xyz[:, 1] = (idx / N) % N; xyz[:, 0] = ((idx / N) / N) % N
    xyz = xyz * (s / (N - 1)) - s / 2
    np.testing.assert_allclose(mesh_util.grid_points(N, s, "cpu", reference_shear=True).numpy(), xyz.astype(np.float32), atol=1e-6)
    np.testing.assert_allclose(mesh_util.grid_points(N, s, "cpu", 7, 19).numpy(), ref[7:19], atol=1e-7)
