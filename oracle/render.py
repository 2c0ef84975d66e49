"""Oracle: ray generation, VolSDF and NeuS volume rendering.

Test infrastructure only (see oracle/__init__.py).  Flat layout: rays are
[R, 3]; outputs are [R, ...] (the reference's batch dim of 1 is dropped).
Results do not depend on ``rayschunk`` when ``perturb=False`` (rays are
independent), so the oracle renders all rays in chunks of its own choosing.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import nets
from .sampling import sdf_to_sigma, fine_sample, sample_pdf


# --------------------------------------------------------------------------
# a1  lift / get_rays  (utils/rend_util.py:95-109, :112-165)
# --------------------------------------------------------------------------
def quat_to_rot(q):
    """(utils/rend_util.py:76-93)"""
    q = F.normalize(q, dim=-1)
    qr, qi, qj, qk = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = torch.ones(*q.shape[:-1], 3, 3)
    R[..., 0, 0] = 1 - 2 * (qj ** 2 + qk ** 2)
    R[..., 0, 1] = 2 * (qj * qi - qk * qr)
    R[..., 0, 2] = 2 * (qi * qk + qr * qj)
    R[..., 1, 0] = 2 * (qj * qi + qk * qr)
    R[..., 1, 1] = 1 - 2 * (qi ** 2 + qk ** 2)
    R[..., 1, 2] = 2 * (qj * qk - qi * qr)
    R[..., 2, 0] = 2 * (qk * qi - qj * qr)
    R[..., 2, 1] = 2 * (qj * qk + qi * qr)
    R[..., 2, 2] = 1 - 2 * (qi ** 2 + qj ** 2)
    return R


def get_rays(c2w, intrinsics, H, W, select_inds=None):
    """c2w [4,4] (or [7] = quaternion + centre), K [4,4] -> rays_o, rays_d [H*W,3] (un-normalised d).

    Pixel (col i, row j) has NO +0.5 offset, ray index = j*W + i (rend_util.py:126-128);
    x = (i - cx + cy*sk/fy - sk*j/fy)/fx, y = (j - cy)/fy, z = 1 (:105-106);
    world = c2w @ [x,y,1,1]; d = world - cam (:157-163).
    """
    if c2w.shape[-1] == 7:
        p = torch.eye(4)
        p[:3, :3] = quat_to_rot(c2w[:4][None])[0]
        p[:3, 3] = c2w[4:]
    else:
        p = c2w
    cam = p[:3, 3]
    jj, ii = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
    i, j = ii.reshape(-1), jj.reshape(-1)
    if select_inds is not None:
        i, j = i[select_inds], j[select_inds]
    fx, fy, cx, cy, sk = intrinsics[0, 0], intrinsics[1, 1], intrinsics[0, 2], intrinsics[1, 2], intrinsics[0, 1]
    z = torch.ones_like(i)
    x_lift = (i - cx + cy * sk / fy - sk * j / fy) / fx * z
    y_lift = (j - cy) / fy * z
    cam_pts = torch.stack([x_lift, y_lift, z, torch.ones_like(z)], dim=0)        # [4, N]
    world = torch.mm(p, cam_pts).transpose(0, 1)[:, :3]
    rays_d = world - cam[None, :]
    rays_o = cam[None, :].expand_as(rays_d)
    return rays_o, rays_d


# a17  near_far_from_sphere  (utils/rend_util.py:168-186)
def near_far_from_sphere(rays_o, rays_d, r=1.0):
    mid = -torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    return (mid - r).clamp_min(0.0), (mid + r).clamp_min(r)


# a24  lin2img  (utils/rend_util.py:238-248)
def lin2img(t, H, W):
    """[H*W, C] -> [C, H, W]"""
    return t.permute(1, 0).reshape(t.shape[1], H, W)


# --------------------------------------------------------------------------
# a16  VolSDF compositing  (models/frameworks/volsdf.py:544-576)
# --------------------------------------------------------------------------
def volsdf_composite(d_all, sigma, radiances, nablas=None, white_bkgd=False):
    delta = d_all[..., 1:] - d_all[..., :-1]
    p = torch.exp(-F.relu(sigma[..., :-1] * delta))
    T = torch.cumprod(torch.cat([torch.ones_like(p[..., :1]), p], dim=-1), dim=-1)[..., :-1]
    tau = (1 - p + 1e-10) * T
    rgb = torch.sum(tau[..., None] * radiances[..., :-1, :], dim=-2)
    depth = torch.sum(tau / (tau.sum(-1, keepdim=True) + 1e-10) * d_all[..., :-1], dim=-1)
    acc = torch.sum(tau, -1)
    if white_bkgd:
        rgb = rgb + (1.0 - acc[..., None])
    out = OrderedDict(rgb=rgb, depth_volume=depth, mask_volume=acc)
    if nablas is not None:
        n = F.normalize(nablas, dim=-1)
        out["normals_volume"] = (n[..., :-1, :] * tau[..., None]).sum(dim=-2)
    out["p_i"], out["visibility_weights"] = p, tau
    return out


# --------------------------------------------------------------------------
# a2/a3/a15  VolSDF volume_render  (volsdf.py:389-615)
# --------------------------------------------------------------------------
def volsdf_render(sd, rays_o, rays_d, near=0.0, far=6.0, obj_bounding_radius=3.0,
                  N_samples=128, N_importance=64, max_upsample_steps=5, max_bisection_steps=10,
                  epsilon=0.1, white_bkgd=False, speed_factor=10.0, multires=6, skips=(4,),
                  rad_multires=-1, rad_multires_view=-1, calc_normal=True, chunk=1024, differentiable=False, u_final=None):
    """u_final [N_rays, N_importance]: perturb=True with the uniform random numbers given (see sampling.fine_sample).
    differentiable=True: the per-sample network queries and the compositing keep their autograd graph w.r.t.
    the tensors of `sd` (sampling stays under no_grad, volsdf.py:479) - Trainer.forward's pass 2."""
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = F.normalize(rays_d.reshape(-1, 3).float(), dim=-1)               # volsdf.py:442
    alpha, beta = nets.volsdf_ab(sd, speed_factor)
    R_bg = obj_bounding_radius
    outs = []
    for c0 in range(0, rays_o.shape[0], chunk):
        o, d = rays_o[c0:c0 + chunk], rays_d[c0:c0 + chunk]
        n = o.shape[0]
        nears = near * torch.ones(n, 1)
        fars = far * torch.ones(n, 1)
        t = torch.linspace(0, 1, N_samples).float()
        d_coarse = nears * (1 - t) + fars * t                                  # volsdf.py:472-474
        t4 = torch.linspace(0, 1, N_samples * 4).float()
        d_init = nears * (1 - t4) + fars * t4                                  # volsdf.py:483-484
        with torch.no_grad():
            d_fine, beta_map, iter_usage = fine_sample(
                lambda x: nets.volsdf_forward_surface(sd, x, R_bg, multires, skips)[0],
                d_init, o, d, alpha, beta, fars, eps=epsilon, max_iter=max_upsample_steps,
                max_bisection=max_bisection_steps, final_N_importance=N_importance,
                N_up=N_samples * 4, det=u_final is None, u_final=None if u_final is None else u_final[c0:c0 + chunk])
        d_all, _ = torch.sort(torch.cat([d_coarse, d_fine], dim=-1), dim=-1)   # volsdf.py:501-502
        pts = o[:, None, :] + d[:, None, :] * d_all[:, :, None]
        v = d[:, None, :].expand_as(pts)
        rad, sdf, nab = nets.volsdf_forward(sd, pts.reshape(-1, 3), v.reshape(-1, 3), R_bg, multires, skips,
                                            rad_multires, rad_multires_view, create_graph=differentiable)
        P = d_all.shape[-1]
        rad, sdf, nab = rad.reshape(n, P, 3), sdf.reshape(n, P), nab.reshape(n, P, 3)
        sigma = sdf_to_sigma(sdf, alpha, beta)
        ret = volsdf_composite(d_all, sigma, rad, nab if calc_normal else None, white_bkgd)
        ret.update(implicit_surface=sdf, implicit_nablas=nab, radiance=rad, alpha=1.0 - ret["p_i"],
                   d_vals=d_all, sigma=sigma, beta_map=beta_map, iter_usage=iter_usage)
        outs.append(ret)
    return OrderedDict((k, torch.cat([o_[k] for o_ in outs], 0)) for k in outs[0])


# --------------------------------------------------------------------------
# a17  NeuS helpers  (models/frameworks/neus.py:29-78)
# --------------------------------------------------------------------------
def cdf_Phi_s(x, s):
    return torch.sigmoid(x * s)


def sdf_to_alpha(sdf, s):
    cdf = cdf_Phi_s(sdf, s)
    a = (cdf[..., :-1] - cdf[..., 1:]) / (cdf[..., :-1] + 1e-10)
    return cdf, torch.clamp_min(a, 0)


def alpha_to_w(alpha):
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[..., :-1]
    return alpha * T


def neus_upsample(sdf_fn, d, o, v, N_importance=64, N_upsample_iters=4, u_new=None):
    """'official_solution' up-sampling  (neus.py:275-303).  u_new [R, N_importance]: perturb=True (det=not perturb,
    neus.py:296) with the uniform numbers given - round i takes columns i * n_new .. (the reference draws them per
    round, rend_util.py:272)."""
    n_new = N_importance // N_upsample_iters
    def query(dv):
        pts = o[:, None, :] + dv[:, :, None] * v[:, None, :]
        return sdf_fn(pts.reshape(-1, 3)).reshape(dv.shape)
    s = query(d)
    for i in range(N_upsample_iters):
        ps, ns = s[..., :-1], s[..., 1:]
        pz, nz = d[..., :-1], d[..., 1:]
        mid = (ps + ns) * 0.5
        slope = (ns - ps) / (nz - pz + 1e-5)
        prev = torch.cat([torch.zeros_like(slope[..., :1]), slope[..., :-1]], dim=-1)
        slope = torch.minimum(prev, slope).clamp(-10.0, 0.0)
        dist = nz - pz
        pe = mid - slope * dist * 0.5
        ne = mid + slope * dist * 0.5
        pc = cdf_Phi_s(pe, 64 * (2 ** i))
        nc = cdf_Phi_s(ne, 64 * (2 ** i))
        a = (pc - nc + 1e-5) / (pc + 1e-5)
        w = alpha_to_w(a)
        d_new = sample_pdf(d, w, n_new, det=u_new is None, u=None if u_new is None else u_new[:, i * n_new:(i + 1) * n_new])
        s_new = query(d_new)
        d = torch.cat([d, d_new], -1)
        s = torch.cat([s, s_new], -1)
        d, order = torch.sort(d, dim=-1, stable=True)
        s = torch.gather(s, -1, order)
    return d, s


def sdf_to_w(sdf, s):
    """neus.py:47-63: visibility weights of the sigmoid-CDF opacity at a fixed s."""
    _, a = sdf_to_alpha(sdf, s)
    return alpha_to_w(a)


def neus_direct_upsample(sdf_fn, d_coarse, near, far, o, v, algo, N_importance=64, N_nograd_samples=2048, fixed_s_recp=1 / 64., u_new=None):
    """'direct_use' (neus.py:242-255: the N_importance fine samples invert the CDF of the COARSE samples' visibility weights at
    s = 1 / fixed_s_recp) and 'direct_more' (:259-269: the same over N_nograd_samples evenly spaced no-gradient samples); both draw
    all N_importance samples at once and sort them in with the coarse ones.  u_new [R, N_importance]: the uniform numbers of perturb=True."""
    def query(dv):
        pts = o[:, None, :] + dv[:, :, None] * v[:, None, :]
        return sdf_fn(pts.reshape(-1, 3)).reshape(dv.shape)
    if algo == "direct_use":
        bins = d_coarse
    elif algo == "direct_more":
        t = torch.linspace(0, 1, N_nograd_samples).float()
        bins = near * (1 - t) + far * t
    else:
        raise ValueError(algo)
    w = sdf_to_w(query(bins), 1.0 / fixed_s_recp)
    d_fine = sample_pdf(bins, w, N_importance, det=u_new is None, u=u_new)
    return torch.sort(torch.cat([d_coarse, d_fine], dim=-1), dim=-1)[0]


# a18  NeuS volume_render  (neus.py:142-424), N_outside=0; upsample_algo 'official_solution' (:275-303), 'direct_use' (:242-255), 'direct_more' (:259-269)
def neus_render(sd, rays_o, rays_d, obj_bounding_radius=1.0, N_samples=64, N_importance=64,
                N_upsample_iters=4, white_bkgd=False, speed_factor=10.0, multires=6, skips=(4,),
                rad_multires=-1, rad_multires_view=4, calc_normal=True, chunk=1024, u_new=None,
                upsample_algo="official_solution", N_nograd_samples=2048, fixed_s_recp=1 / 64.):
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = F.normalize(rays_d.reshape(-1, 3).float(), dim=-1)
    outs = []
    for c0 in range(0, rays_o.shape[0], chunk):
        o, v = rays_o[c0:c0 + chunk], rays_d[c0:c0 + chunk]
        n = o.shape[0]
        near, far = near_far_from_sphere(o, v, r=obj_bounding_radius)
        t = torch.linspace(0, 1, N_samples).float()
        d_coarse = near * (1 - t) + far * t
        sdf_fn = lambda x: nets.surface_forward(sd, x, multires, skips)[0]
        u_c = None if u_new is None else u_new[c0:c0 + chunk]
        with torch.no_grad():
            if upsample_algo == "official_solution":
                d_all, _ = neus_upsample(sdf_fn, d_coarse, o, v, N_importance, N_upsample_iters, u_c)
            else:
                d_all = neus_direct_upsample(sdf_fn, d_coarse, near, far, o, v, upsample_algo, N_importance, N_nograd_samples, fixed_s_recp, u_c)
        pts = o[:, None, :] + v[:, None, :] * d_all[:, :, None]
        d_mid = 0.5 * (d_all[..., 1:] + d_all[..., :-1])
        pts_mid = o[:, None, :] + v[:, None, :] * d_mid[:, :, None]
        P = d_all.shape[-1]
        sdf, nab, _ = nets.surface_forward_with_nablas(sd, pts.reshape(-1, 3), multires, skips)
        sdf, nab = sdf.reshape(n, P), nab.reshape(n, P, 3)
        cdf, alpha = sdf_to_alpha(sdf, nets.neus_s(sd, speed_factor))
        rad = nets.neus_forward_radiance(sd, pts_mid.reshape(-1, 3), v[:, None, :].expand_as(pts_mid).reshape(-1, 3),
                                         multires, skips, rad_multires, rad_multires_view).reshape(n, P - 1, 3)
        w = alpha_to_w(alpha)
        rgb = torch.sum(w[..., None] * rad, -2)
        depth = torch.sum(w / (w.sum(-1, keepdim=True) + 1e-10) * d_mid, -1)
        acc = torch.sum(w, -1)
        if white_bkgd:
            rgb = rgb + (1.0 - acc[..., None])
        ret = OrderedDict(rgb=rgb, depth_volume=depth, mask_volume=acc)
        if calc_normal:
            nn_ = F.normalize(nab, dim=-1)
            ret["normals_volume"] = (nn_[..., :P - 1, :] * w[..., None]).sum(dim=-2)
        ret.update(implicit_nablas=nab, implicit_surface=sdf, radiance=rad, alpha=alpha, cdf=cdf,
                   visibility_weights=w, d_final=d_mid, d_all=d_all)
        outs.append(ret)
    return OrderedDict((k, torch.cat([o_[k] for o_ in outs], 0)) for k in outs[0])
