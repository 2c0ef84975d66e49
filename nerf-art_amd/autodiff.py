"""Differentiable evaluation of the per-sample networks and the compositing - the part of the fine-tune step
that needs parameter gradients (reference Trainer.forward pass 2, models/frameworks/volsdf.py:753-770,
neus.py:520-576).

STATUS: this is the *library* path of row a19 - PyTorch autograd, GEMMs on rocBLAS
through torch, including the double backward through the SDF net that the eikonal term and the normal input of
the radiance net need.  Everything that does not need gradients in a training step (pass 1, and the sampling of
pass 2: 512 x (1 + rounds) SDF evaluations per ray) runs on the hand-written HIP kernels.  The functions here
follow the reference formulas line by line so that the hand-written backward kernels that replace them can be
checked against them on the GPU at full size.
"""
import numpy as np
import torch
import torch.nn.functional as F


def embed(x: torch.Tensor, multires: int) -> torch.Tensor:
    """models/base.py:46-64: cat[x, sin(2^k x), cos(2^k x), ...], k = 0..multires-1; identity for multires = -1."""
    if multires < 0:
        return x
    out = [x]
    for k in range(multires):
        f = 2.0 ** k
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, dim=-1)


def wn_linear(layer, h: torch.Tensor) -> torch.Tensor:
    """nn.utils.weight_norm(nn.Linear): w = g * v / ||v||_row (base.py:226-227)."""
    return F.linear(h, torch._weight_norm(layer.weight_v, layer.weight_g, 0), layer.bias)


def softplus100(x):
    return F.softplus(x, beta=100.0, threshold=20.0)


def surface_forward(surf, x: torch.Tensor):
    """ImplicitSurface.forward (base.py:243-263) -> (sdf, feat)."""
    e = embed(x, surf.embed_multires)
    h = e
    for i in range(surf.D):
        if i in surf.skips:
            h = torch.cat([h, e], dim=-1) / np.sqrt(2)
        h = softplus100(wn_linear(surf.surface_fc_layers[i], h))
    out = wn_linear(surf.surface_fc_layers[surf.D], h)
    return out[..., 0], out[..., 1:]


def surface_forward_with_nablas(surf, x: torch.Tensor):
    """ImplicitSurface.forward_with_nablas under grad mode (base.py:265-282): nabla keeps its graph."""
    with torch.enable_grad():
        xg = x.detach().requires_grad_(True)
        sdf, feat = surface_forward(surf, xg)
        nabla = torch.autograd.grad(sdf, xg, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0]
    return sdf, nabla, feat


def radiance_forward(rad, x, view_dirs, normals, feat):
    """RadianceNet.forward (base.py:372-391)."""
    h = torch.cat([embed(x, rad.embed_multires), embed(view_dirs, rad.embed_multires_view), normals, feat], dim=-1)
    for i in range(rad.D):
        h = F.relu(wn_linear(rad.layers[i], h))
    return torch.sigmoid(wn_linear(rad.layers[rad.D], h))


def volsdf_point_forward(model, x, view_dirs):
    """VolSDF.forward (volsdf.py:349-370): sphere clamp on sdf only, raw nabla into the radiance net."""
    sdf, nabla, feat = surface_forward_with_nablas(model.implicit_surface, x)
    d_bg = model.obj_bounding_radius - x.norm(dim=-1)
    sdf = torch.where(d_bg < sdf, d_bg, sdf)
    return radiance_forward(model.radiance_net, x, view_dirs, nabla, feat), sdf, nabla


def sdf_to_sigma(sdf, alpha, beta):
    """volsdf.py:34-53."""
    exp = 0.5 * torch.exp(-torch.abs(sdf) / beta)
    psi = torch.where(sdf >= 0, exp, 1 - exp)
    return alpha * psi


def volsdf_composite(d_all, sigma, radiances, nablas=None, white_bkgd=False):
    """volsdf.py:544-576 (last sample dropped)."""
    delta = d_all[..., 1:] - d_all[..., :-1]
    p_i = torch.exp(-F.relu(sigma[..., :-1] * delta))
    shifted = torch.cat([torch.ones_like(p_i[..., :1]), p_i], dim=-1)
    tau = (1 - p_i + 1e-10) * torch.cumprod(shifted, dim=-1)[..., :-1]
    rgb = torch.sum(tau[..., None] * radiances[..., :-1, :], dim=-2)
    acc = torch.sum(tau, -1)
    depth = torch.sum(tau / (acc[..., None] + 1e-10) * d_all[..., :-1], -1)
    if white_bkgd:
        rgb = rgb + (1.0 - acc[..., None])
    out = {"rgb": rgb, "depth_volume": depth, "mask_volume": acc}
    if nablas is not None:
        out["normals_volume"] = (F.normalize(nablas[..., :-1, :], dim=-1) * tau[..., None]).sum(dim=-2)
    return out


class CompositeRGB(torch.autograd.Function):
    """rgb = composite(d_all, sigma(sdf; alpha, beta), radiance) with the hand-written HIP forward and backward
    kernels (nerfart_volsdf_composite / _composite_bwd): the per-ray stage of pass 2 without an autograd graph."""

    @staticmethod
    def forward(ctx, d_all, sdf, radiance, alpha, beta, white_bkgd):
        from . import hip
        d_all, sdf, radiance = d_all.contiguous(), sdf.contiguous(), radiance.contiguous()
        rgb, _, _ = hip.volsdf_composite(d_all, sdf, radiance, float(alpha), float(beta), white_bkgd)
        ctx.save_for_backward(d_all, sdf, radiance, alpha, beta)
        ctx.white_bkgd = white_bkgd
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        from . import hip
        d_all, sdf, radiance, alpha, beta = ctx.saved_tensors
        g_sdf, g_rad, g_ab = hip.volsdf_composite_bwd(d_all, sdf, radiance, float(alpha), float(beta), g_rgb.contiguous(), ctx.white_bkgd)
        return None, g_sdf, g_rad, g_ab[0].reshape(alpha.shape), g_ab[1].reshape(beta.shape), None


def volsdf_render_samples(model, rays_o, rays_dn, d_all, white_bkgd=False, calc_normal=True, native_composite=None):
    """Differentiable part of volume_render for given (non-differentiable) sample depths d_all [R, P]:
    rays_o / rays_dn [R, 3] (directions normalised).  Returns the reference's extras incl. implicit_nablas."""
    R, P = d_all.shape
    pts = rays_o[:, None, :] + rays_dn[:, None, :] * d_all[:, :, None]
    v = rays_dn[:, None, :].expand_as(pts)
    rad, sdf, nab = volsdf_point_forward(model, pts.reshape(-1, 3), v.reshape(-1, 3))
    rad, sdf, nab = rad.reshape(R, P, 3), sdf.reshape(R, P), nab.reshape(R, P, 3)
    alpha, beta = model.forward_ab()
    if native_composite is None:
        native_composite = d_all.is_cuda
    if native_composite:
        # only rgb is produced (all the fine-tune losses need); depth / mask / normals maps come from pass 1
        out = {"rgb": CompositeRGB.apply(d_all, sdf, rad, alpha, beta, white_bkgd)}
        out.update(implicit_surface=sdf, implicit_nablas=nab, radiance=rad, d_vals=d_all)
        return out
    sigma = sdf_to_sigma(sdf, alpha, beta)
    out = volsdf_composite(d_all, sigma, rad, nab if calc_normal else None, white_bkgd)
    out.update(implicit_surface=sdf, implicit_nablas=nab, radiance=rad, sigma=sigma, d_vals=d_all)
    return out


# ---- NeuS (models/frameworks/neus.py:29-78, 310-395) -------------------------------------------------------
def cdf_Phi_s(x, s):
    return torch.sigmoid(x * s)


def sdf_to_alpha(sdf, s):
    cdf = cdf_Phi_s(sdf, s)
    opacity_alpha = (cdf[..., :-1] - cdf[..., 1:]) / (cdf[..., :-1] + 1e-10)
    return cdf, torch.clamp_min(opacity_alpha, 0)


def alpha_to_w(alpha):
    shifted = torch.cat([torch.ones_like(alpha[..., :1]), 1.0 - alpha + 1e-10], dim=-1)
    return alpha * torch.cumprod(shifted, dim=-1)[..., :-1]


def neus_render_samples(model, rays_o, rays_dn, d_all, white_bkgd=False, calc_normal=True):
    """neus.py:310-395 for given depths d_all [R, P]: SDF + nablas at the P samples, radiance at the P-1 mid points."""
    R, P = d_all.shape
    pts = rays_o[:, None, :] + rays_dn[:, None, :] * d_all[:, :, None]
    d_mid = 0.5 * (d_all[..., 1:] + d_all[..., :-1])
    pts_mid = rays_o[:, None, :] + rays_dn[:, None, :] * d_mid[:, :, None]
    sdf, nab, _ = surface_forward_with_nablas(model.implicit_surface, pts.reshape(-1, 3))
    sdf, nab = sdf.reshape(R, P), nab.reshape(R, P, 3)
    cdf, alpha = sdf_to_alpha(sdf, model.forward_s())
    xm = pts_mid.reshape(-1, 3)
    _, nab_m, feat_m = surface_forward_with_nablas(model.implicit_surface, xm)
    vm = rays_dn[:, None, :].expand_as(pts_mid).reshape(-1, 3)
    rad = radiance_forward(model.radiance_net, xm, vm, nab_m, feat_m).reshape(R, P - 1, 3)
    w = alpha_to_w(alpha)
    rgb = torch.sum(w[..., None] * rad, dim=-2)
    acc = torch.sum(w, -1)
    depth = torch.sum(w / (acc[..., None] + 1e-10) * d_mid, -1)
    if white_bkgd:
        rgb = rgb + (1.0 - acc[..., None])
    out = {"rgb": rgb, "depth_volume": depth, "mask_volume": acc, "implicit_surface": sdf, "implicit_nablas": nab,
           "radiance": rad, "alpha": alpha, "cdf": cdf, "visibility_weights": w, "d_final": d_all}
    if calc_normal:
        out["normals_volume"] = (F.normalize(nab[..., :-1, :], dim=-1) * w[..., None]).sum(dim=-2)
    return out
