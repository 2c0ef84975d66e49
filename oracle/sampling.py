"""Oracle: VolSDF density, error bound, inverse-CDF samplers and Algorithm 1 (fine_sample).

Test infrastructure only (see oracle/__init__.py).  Flat ray layout: every
per-ray tensor is [R, ...]; the reference's leading batch dim of 1 is dropped
(it only ever broadcasts).  The upsampling loop keeps an explicit list of
active ray indices instead of the reference's boolean masks-of-masks.
"""
import numpy as np
import torch


# a11  sdf_to_sigma  (models/frameworks/volsdf.py:34-53)
def sdf_to_sigma(sdf, alpha, beta):
    """sigma = alpha * (0.5 exp(-|s|/beta) if s >= 0 else 1 - 0.5 exp(-|s|/beta))."""
    e = 0.5 * torch.exp(-torch.abs(sdf) / beta)
    return alpha * torch.where(sdf >= 0, e, 1 - e)


def _opacity_R(d, sdf, alpha, beta):
    """R_t[k] = sum_{i<k} sigma_i * delta_i, k = 0..N-2  (volsdf.py:75-81, :127-132)."""
    sigma = sdf_to_sigma(sdf, alpha, beta)
    delta = d[..., 1:] - d[..., :-1]
    return torch.cat([torch.zeros_like(d[..., :1]), torch.cumsum(sigma[..., :-1] * delta, dim=-1)], dim=-1)[..., :-1]


# a12  error_bound  (volsdf.py:56-94)
def error_bound(d, sdf, alpha, beta):
    """[..., N] -> [..., N-1] opacity error bound of each interval; NaN (0*inf) -> +inf."""
    delta = d[..., 1:] - d[..., :-1]
    R_t = _opacity_R(d, sdf, alpha, beta)
    a = torch.abs(sdf)
    d_star = torch.clamp_min(0.5 * (a[..., :-1] + a[..., 1:] - delta), 0.0)
    err = alpha / (4 * beta) * (delta ** 2) * torch.exp(-d_star / beta)
    bounds = torch.exp(-R_t) * (torch.exp(torch.cumsum(err, dim=-1)) - 1.0)
    bounds[torch.isnan(bounds)] = np.inf
    return bounds


def _invert_cdf(bins, cdf, n, det=True, u=None, eps=1e-5):
    """Piece-wise linear inverse CDF shared by sample_pdf / sample_cdf
    (utils/rend_util.py:267-293, :302-328): lower-bound search, clamp the bracket to the
    array, guard a < eps denominator with 1."""
    if u is None:
        if det:
            u = torch.linspace(0.0, 1.0, steps=n).expand(*cdf.shape[:-1], n)
        else:
            u = torch.rand(*cdf.shape[:-1], n)
    u = u.contiguous()
    idx = torch.searchsorted(cdf.contiguous(), u, right=False)
    lo = torch.clamp_min(idx - 1, 0)
    hi = torch.clamp_max(idx, cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    denom = c_hi - c_lo
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    t = (u - c_lo) / denom
    return b_lo + t * (b_hi - b_lo)


# a14  sample_pdf  (utils/rend_util.py:256-293)
def sample_pdf(bins, weights, n, det=True, u=None):
    """bins[..., N], weights[..., N-1] -> [..., n]."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    return _invert_cdf(bins, cdf, n, det, u)


# a14  sample_cdf  (utils/rend_util.py:295-328)
def sample_cdf(bins, cdf, n, det=True, u=None):
    """bins[..., N], cdf[..., N-1] (a leading 0 is prepended) -> [..., n]."""
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    return _invert_cdf(bins, cdf, n, det, u)


def opacity_invert_cdf_sample(d, sdf, alpha, beta, n, det=True, u=None):
    """Inverse-CDF samples of the opacity 1 - exp(-R_t)  (volsdf.py:122-136).  u [..., n]: the uniform numbers
    sample_cdf(det=False) would draw (rend_util.py:306-307), supplied so that a run is reproducible."""
    return sample_cdf(d, 1 - torch.exp(-_opacity_R(d, sdf, alpha, beta)), n, det=det, u=u)


def _merge_sorted(d_old, s_old, d_new, s_new):
    """cat + (stable) sort by depth + gather sdf  (volsdf.py:217-228)."""
    d = torch.cat([d_old, d_new], -1)
    s = torch.cat([s_old, s_new], -1)
    d, order = torch.sort(d, dim=-1, stable=True)
    return d, torch.gather(s, -1, order)


# a13  fine_sample  (volsdf.py:97-302) - VolSDF Algorithm 1
def fine_sample(sdf_fn, d_init, rays_o, rays_d, alpha_net, beta_net, far,
                eps=0.1, max_iter=5, max_bisection=10, final_N_importance=64, N_up=128, det=True, u_final=None):
    """
    u_final [R, final_N]: perturb=True (det = not perturb, volsdf.py:122) with the uniform random numbers of every
    ray given (the reference draws them per converged subset, rend_util.py:307: same distribution, not reproducible).
    sdf_fn(pts[M,3]) -> sdf[M]      (VolSDF.forward_surface incl. the sphere clamp)
    d_init[R, N0], rays_o/rays_d[R,3], alpha_net/beta_net scalar tensors, far scalar or [R,1]
    returns d_fine[R, final_N], beta_map[R, 1], iter_usage[R] (0..max_iter, -1 = never converged)
    """
    R, N0 = d_init.shape

    def query(dv, idx):
        pts = rays_o[idx, None, :] + rays_d[idx, None, :] * dv[..., :, None]
        return sdf_fn(pts.reshape(-1, 3)).reshape(dv.shape)

    all_idx = torch.arange(R)
    if not isinstance(far, torch.Tensor):
        far = far * torch.ones(R, 1)
    beta = torch.sqrt((far ** 2) / (4 * (N0 - 1) * np.log(1 + eps)))       # beta_+ init (volsdf.py:149)
    d_fine = torch.zeros(R, final_N_importance)
    usage = torch.zeros(R)
    converged = torch.zeros(R, dtype=torch.bool)

    d_all, s_all = d_init.clone(), query(d_init, all_idx)
    net_max = error_bound(d_all, s_all, alpha_net, beta_net).max(dim=-1).values
    need = net_max > eps
    act = all_idx[need]
    done = all_idx[~need]
    if done.numel() > 0:
        d_fine[done] = opacity_invert_cdf_sample(d_all[done], s_all[done], alpha_net, beta_net,
                                                 final_N_importance, det, None if u_final is None else u_final[done])
        converged[done] = True
    d_act, s_act = d_all[act], s_all[act]
    b_act = error_bound(d_act, s_act, 1.0 / beta[act], beta[act])

    it = 0
    while it < max_iter and act.numel() > 0:
        it += 1
        # upsample proportional to the current bound; det=True, drop the two end points (volsdf.py:196)
        d_new = sample_pdf(d_act, b_act, N_up + 2, det=True)[..., 1:-1]
        s_new = query(d_new, act)
        d_act, s_act = _merge_sorted(d_act, s_act, d_new, s_new)
        nm = error_bound(d_act, s_act, alpha_net, beta_net).max(dim=-1).values
        ok = nm <= eps                       # reference: sub_mask = net_bounds_max > eps
        if ok.any():
            fin = act[ok]
            d_fine[fin] = opacity_invert_cdf_sample(d_act[ok], s_act[ok], alpha_net, beta_net,
                                                    final_N_importance, det, None if u_final is None else u_final[fin])
            usage[fin] = it
            converged[fin] = True
        keep = ~ok
        if not keep.any():
            act = act[keep]
            break
        act, d_act, s_act = act[keep], d_act[keep], s_act[keep]
        # bisection for beta_+ with B(beta_+) == eps (volsdf.py:260-275)
        b_hi = beta[act].clone()
        b_lo = beta_net * torch.ones_like(b_hi)
        for _ in range(max_bisection):
            b_mid = 0.5 * (b_lo + b_hi)
            m = error_bound(d_act, s_act, 1.0 / b_mid, b_mid).max(dim=-1).values
            le = (m <= eps)[:, None]
            b_hi = torch.where(le, b_mid, b_hi)
            b_lo = torch.where(~le, b_mid, b_lo)
        beta[act] = b_hi
        b_act = torch.clamp(error_bound(d_act, s_act, 1.0 / beta[act], beta[act]), 0, 1e5)

    if act.numel() > 0:                      # never converged: sample with the last beta_+ (volsdf.py:294-300)
        bp = beta[act]
        d_fine[act] = opacity_invert_cdf_sample(d_act, s_act, 1.0 / bp, bp, final_N_importance, det,
                                                None if u_final is None else u_final[act])
        usage[act] = -1
    beta[converged] = beta_net
    return d_fine, beta, usage
