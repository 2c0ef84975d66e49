"""Oracle: surface ray casting (reference models/ray_casting.py) restated on a state dict - TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  Pinned by tests/golden/raycast_golden.npz (captured from the reference's functions)."""
import torch
import torch.nn.functional as F

from . import nets


def _sdf(sd, x):
    return nets.surface_forward(sd, x)[0]


def root_finding(sd, rays_o, rays_dn, near=0.0, far=6.0, n_steps=256, logit_tau=0.0, n_secant=8, fill_inf=True):
    """rays [R, 3] (dn normalised) -> (depth [R], pts [R, 3], mask, mask_sign_change)   (ray_casting.py:35-160)."""
    R = rays_o.shape[0]
    with torch.no_grad():
        t = torch.linspace(0.0, 1.0, n_steps)
        near_t = near if torch.is_tensor(near) else torch.full((R,), float(near))
        far_t = far if torch.is_tensor(far) else torch.full((R,), float(far))
        d = near_t[:, None] * (1 - t) + far_t[:, None] * t                                     # :74
        val = _sdf(sd, rays_o[:, None, :] + d[..., None] * rays_dn[:, None, :]) - logit_tau    # :77-86
        starts_outside = val[:, 0] > 0                                                          # :90
        crossing = torch.sign(val[:, :-1] * val[:, 1:]) < 0                                     # the negative entries of the cost matrix :93-100
        has = crossing.any(dim=1)                                                               # :104
        first = torch.where(has, crossing.float().argmax(dim=1), torch.zeros(R, dtype=torch.long))
        rows = torch.arange(R)
        mask = has & (val[rows, first] > 0) & starts_outside                                    # :107-109
        nxt = torch.clamp(first + 1, max=n_steps - 1)
        d_high, f_high, d_low, f_low = d[rows, first][mask], val[rows, first][mask], d[rows, nxt][mask], val[rows, nxt][mask]
        o_m, dn_m = rays_o[mask], rays_dn[mask]
        d_pred = -f_low * (d_high - d_low) / (f_high - f_low) + d_low                           # :15
        for _ in range(n_secant):                                                               # :16-29
            f_mid = _sdf(sd, o_m + d_pred[:, None] * dn_m) - logit_tau
            low = f_mid < 0
            d_low, f_low = torch.where(low, d_pred, d_low), torch.where(low, f_mid, f_low)
            d_high, f_high = torch.where(low, d_high, d_pred), torch.where(low, f_high, f_mid)
            d_pred = -f_low * (d_high - d_low) / (f_high - f_low) + d_low
        pts = torch.ones(R, 3)
        pts[mask] = o_m + d_pred[:, None] * dn_m
        depth = torch.full((R,), float("inf")) if fill_inf else far_t.clone()
        depth[mask] = d_pred
        depth[~starts_outside] = 0.0                                                            # :150
    return depth, pts, mask, has


def sphere_tracing(sd, rays_o, rays_dn, near=0.0, far=6.0, n_iters=20):
    """(ray_casting.py:163-182)"""
    with torch.no_grad():
        d = torch.full((rays_o.shape[0],), float(near))
        live = torch.ones_like(d, dtype=torch.bool)
        for _ in range(n_iters):
            s = _sdf(sd, rays_o + rays_dn * d[:, None])
            d = torch.where(live, d + s, d)
            live = live & ~(d > far) & ~(d < 0)
        return d, rays_o + rays_dn * d[:, None], live


def surface_render(sd, rays_o, rays_d, algo, rad_multires_view=-1, **cfgs):
    """(ray_casting.py:185-263) -> dict(rgb, depth, implicit_nablas, mask_surface, normals_surface)."""
    dn = F.normalize(rays_d, dim=-1)
    if algo == "root_finding":
        depth, pts, mask, _ = root_finding(sd, rays_o, dn, **cfgs)
    else:
        depth, pts, mask = sphere_tracing(sd, rays_o, dn, **cfgs)
    sdf, nab, feat = nets.surface_forward_with_nablas(sd, pts)
    rgb = nets.radiance_forward(sd, pts, dn, nab, feat, -1, rad_multires_view)
    normals = F.normalize(nab, dim=-1)
    return {"rgb": torch.where(mask[:, None], rgb, torch.zeros_like(rgb)), "depth": depth, "implicit_nablas": nab, "mask_surface": mask,
            "normals_surface": torch.where(mask[:, None], normals, torch.zeros_like(normals))}
