"""Surface renderer: the call shapes of the reference's models/ray_casting.py - `root_finding_surface_points`
(:35-160), `sphere_tracing_surface_points` (:163-182), `surface_render` (:185-263; render.py --use_surface_render) - on the HIP
kernels: the marched / refined points are evaluated by the SDF kernel K2 straight from rays + depths (never materialised),
the per-ray analysis runs in csrc/ray_casting.hip (first sign change, secant bracket updates, sphere-trace step).

`surface_query_fn` is the model's `implicit_surface` (as the reference passes it): its weights are read from the owning
model's packed blob; any other callable is rejected (there is no eager path).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import hip
from .nets import ImplicitSurface

_u8 = torch.uint8


def _blob_of(surface_query_fn):
    if not isinstance(surface_query_fn, ImplicitSurface):
        raise TypeError("ray casting runs on the HIP SDF kernel: pass model.implicit_surface (a nerfart_amd ImplicitSurface)")
    m = surface_query_fn._model()
    return m.packed()[0], m.precision_id


def _flat_rays(rays_o, rays_d, batched):
    if not batched:
        rays_o, rays_d = rays_o[None], rays_d[None]
    B, N = rays_o.shape[0], rays_o.shape[-2]
    return rays_o.reshape(-1, 3).float().contiguous(), rays_d.reshape(-1, 3).float().contiguous(), B, N


def _per_ray(v, shape, device):
    """near / far: a number or a [(B), N_rays] tensor -> (tensor [R] or None, scalar)."""
    if isinstance(v, torch.Tensor):
        return v.to(device).float().expand(shape).reshape(-1).contiguous(), 0.0
    return None, float(v)


def _sdf_at(blob, prec, o, dn, depth_col):
    return hip.sdf_fwd_rays(blob, o, dn, depth_col, 0.0, precision=prec)


def root_finding_surface_points(surface_query_fn, rays_o, rays_d, near=0.0, far=6.0, batched=True, batched_info={}, N_steps=256,
                                logit_tau=0.0, method="secant", N_secant_steps=8, fill_inf=True):
    """rays_d already normalised.  -> (d_pred_out [(B), N], pt_pred [(B), N, 3], mask, mask_sign_change) as ray_casting.py:35-160:
    march N_steps depths between near and far, take the first sign change of (sdf - logit_tau) if it goes from outside to inside
    and the ray starts outside, refine with N_secant_steps secant iterations; depth inf (fill_inf) / far where nothing is hit,
    0 where the ray starts inside; pt_pred = 1 where mask is false."""
    blob, prec = _blob_of(surface_query_fn)
    o, dn, B, N = _flat_rays(rays_o, rays_d, batched)
    R, dev = o.shape[0], o.device
    near_t, near_s = _per_ray(near, (B, N), dev)
    far_t, far_s = _per_ray(far, (B, N), dev)
    with torch.no_grad():
        t = hip.lin_table(N_steps, dev)
        depth = torch.empty(R, N_steps, device=dev)
        hip._check(hip.lib.nerfart_linspace_depths(hip._dev(t), N_steps, hip._dev(near_t), hip._dev(far_t), near_s, far_s, R, hip._dev(depth),
                                                   N_steps, hip._stream()), "nerfart_linspace_depths")
        val = _sdf_at(blob, prec, o, dn, depth)
        mask, msc, m0 = (torch.empty(R, dtype=_u8, device=dev) for _ in range(3))
        brk, d_pred = torch.empty(R, 4, device=dev), torch.empty(R, device=dev)
        hip._check(hip.lib.nerfart_first_crossing(hip._dev(val), hip._dev(depth), R, N_steps, float(logit_tau), hip._dev(mask, _u8), hip._dev(msc, _u8),
                                                  hip._dev(m0, _u8), hip._dev(brk), hip._dev(d_pred), hip._stream()), "nerfart_first_crossing")
        if method == "secant":
            for _ in range(N_secant_steps):
                f_mid = _sdf_at(blob, prec, o, dn, d_pred[:, None].contiguous())
                hip._check(hip.lib.nerfart_secant_update(hip._dev(f_mid.reshape(-1)), R, float(logit_tau), hip._dev(mask, _u8), hip._dev(brk),
                                                         hip._dev(d_pred), hip._stream()), "nerfart_secant_update")
        else:
            d_pred = torch.ones(R, device=dev)
        d_out, pt = torch.empty(R, device=dev), torch.empty(R, 3, device=dev)
        hip._check(hip.lib.nerfart_root_finish(hip._dev(o), hip._dev(dn), R, hip._dev(mask, _u8), hip._dev(m0, _u8), hip._dev(d_pred), hip._dev(far_t),
                                               far_s, int(bool(fill_inf)), hip._dev(d_out), hip._dev(pt), hip._stream()), "nerfart_root_finish")
    shp = (B, N) if batched else (N,)
    return d_out.reshape(shp), pt.reshape(*shp, 3), mask.bool().reshape(shp), msc.bool().reshape(shp)


def sphere_tracing_surface_points(implicit_surface, rays_o, rays_d, near=0.0, far=6.0, batched=True, batched_info={}, N_iters=20):
    """-> (d_preds, pts, mask): N_iters steps d += sdf(o + d dir) on rays still inside [0, far] (ray_casting.py:163-182)."""
    blob, prec = _blob_of(implicit_surface)
    o, dn, B, N = _flat_rays(rays_o, rays_d, batched)
    R, dev = o.shape[0], o.device
    near_t, near_s = _per_ray(near, (B, N), dev)
    far_t, far_s = _per_ray(far, (B, N), dev)
    with torch.no_grad():
        d = near_t.clone() if near_t is not None else torch.full((R,), near_s, device=dev)
        mask = torch.ones(R, dtype=_u8, device=dev)
        for _ in range(N_iters):
            sdf = _sdf_at(blob, prec, o, dn, d[:, None].contiguous())
            hip._check(hip.lib.nerfart_sphere_trace_step(hip._dev(sdf.reshape(-1)), R, hip._dev(far_t), far_s, hip._dev(d), hip._dev(mask, _u8),
                                                         hip._stream()), "nerfart_sphere_trace_step")
        pts = o + dn * d[:, None]
    shp = (B, N) if batched else (N,)
    return d.reshape(shp), pts.reshape(*shp, 3), mask.bool().reshape(shp)


def surface_render(rays_o, rays_d, model, calc_normal=True, rayschunk=8192, netchunk=1048576, batched=True, use_view_dirs=True,
                   show_progress=False, ray_casting_algo="", ray_casting_cfgs={}, **not_used_kwargs):
    """render.py's `--use_surface_render` path (ray_casting.py:185-263): ray cast to the surface, shade the hit points with
    model.forward.  -> (colors [(B), N, 3] (0 where nothing is hit), depths, extras{implicit_nablas, mask_surface[, normals_surface]}).
    rays_d is NOT normalised on entry.  Rays are marched in slices of rayschunk (results do not depend on it); netchunk is accepted and ignored."""
    if ray_casting_algo not in ("root_finding", "sphere_tracing"):
        raise NotImplementedError(f"ray_casting_algo {ray_casting_algo!r}")
    if not use_view_dirs:
        raise NotImplementedError("use_view_dirs=False is not used by any reference config")
    with torch.no_grad():
        shape = [rays_d.shape[0], -1, 3] if batched else [-1, 3]
        o = rays_o.reshape(shape).float()
        dn = F.normalize(rays_d.reshape(shape).float(), dim=-1)
        # rays are independent: march them in slices of `rayschunk` (never more than 2^31 march samples per launch), as the
        # reference does (ray_casting.py:241-262) - the [rays, N_steps] march buffers stay bounded, the results are identical
        n_steps = int(ray_casting_cfgs.get("N_steps", 256)) if ray_casting_algo == "root_finding" else 1
        N = o.shape[-2]
        step = max(1, min(int(rayschunk) if rayschunk else N, ((1 << 31) - 1) // max(n_steps, 1)))
        parts = []
        for s in range(0, max(N, 1), step):                             # N == 0 (an empty ray batch): one empty slice, the kernels' n <= 0 path
            oc, dc = o[..., s:s + step, :].contiguous(), dn[..., s:s + step, :].contiguous()
            cfg = {k: (v[..., s:s + step] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[-1] == N else v) for k, v in ray_casting_cfgs.items()}
            if ray_casting_algo == "root_finding":
                depths, pts, mask, _ = root_finding_surface_points(model.implicit_surface, oc, dc, batched=batched, **cfg)
            else:
                depths, pts, mask = sphere_tracing_surface_points(model.implicit_surface, oc, dc, batched=batched, **cfg)
            colors, _, nablas = model.forward(pts.contiguous(), dc)
            parts.append((torch.where(mask[..., None], colors, torch.zeros_like(colors)), depths, nablas, mask))
        colors, depths, nablas, mask = (torch.cat([p[i] for p in parts], dim=-2 if i in (0, 2) else -1) for i in range(4))
        extras = OrderedDict([("implicit_nablas", nablas), ("mask_surface", mask)])
        if calc_normal:
            normals = F.normalize(nablas, dim=-1)
            extras["normals_surface"] = torch.where(mask[..., None], normals, torch.zeros_like(normals))
    return colors, depths, extras
