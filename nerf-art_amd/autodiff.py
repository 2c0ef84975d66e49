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


def surface_forward(surf, x: torch.Tensor, return_h: bool = False):
    """ImplicitSurface.forward (base.py:243-263) -> (sdf, feat)  [return_h: (sdf, h7) - the geometry feature is then
    left to the radiance kernels, which own the rows 1.. of the last linear layer]."""
    e = embed(x, surf.embed_multires)
    h = e
    for i in range(surf.D):
        if i in surf.skips:
            h = torch.cat([h, e], dim=-1) / np.sqrt(2)
        h = softplus100(wn_linear(surf.surface_fc_layers[i], h))
    last = surf.surface_fc_layers[surf.D]
    if return_h:
        w = torch._weight_norm(last.weight_v, last.weight_g, 0)
        return F.linear(h, w[:1], last.bias[:1])[..., 0], h
    out = wn_linear(last, h)
    return out[..., 0], out[..., 1:]


def surface_forward_with_nablas(surf, x: torch.Tensor, return_h: bool = False):
    """ImplicitSurface.forward_with_nablas under grad mode (base.py:265-282): nabla keeps its graph."""
    with torch.enable_grad():
        xg = x.detach().requires_grad_(True)
        sdf, feat = surface_forward(surf, xg, return_h)
        nabla = torch.autograd.grad(sdf, xg, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0]
    return sdf, nabla, feat


def radiance_forward(rad, x, view_dirs, normals, feat):
    """RadianceNet.forward (base.py:372-391)."""
    h = torch.cat([embed(x, rad.embed_multires), embed(view_dirs, rad.embed_multires_view), normals, feat], dim=-1)
    for i in range(rad.D):
        h = F.relu(wn_linear(rad.layers[i], h))
    return torch.sigmoid(wn_linear(rad.layers[rad.D], h))


# ---- radiance net on the hand-written kernels (forward with activation dumps, backward chain, GEMM operands) ----
_RAD_DUMP_PER_TILE = 5 * 8 * 8 * 1024
_UNIT_PERM = None


def _unit_perm(device):
    """column c = (unit * 4 + lane group) * 8 + e of a dumped matrix -> natural feature index."""
    global _UNIT_PERM
    if _UNIT_PERM is None:
        from .packing import unit_feature_hidden
        _UNIT_PERM = torch.tensor([unit_feature_hidden(u, g, e) for u in range(8) for g in range(4) for e in range(8)])
    return _UNIT_PERM.to(device)


def _dump_matrix(dump: torch.Tensor, slot: int, M: int) -> torch.Tensor:
    """[M, 256] fp32 (columns in unit order) of one dumped activation / delta."""
    T = dump.numel() // _RAD_DUMP_PER_TILE
    v = dump.view(torch.bfloat16).view(T, 5, 8, 8, 4, 16, 8)[:, slot]            # tile, unit, wave, g, j, e
    return v.permute(0, 2, 4, 1, 3, 5).reshape(T * 128, 256)[:M].float()


class RadianceNetFn(torch.autograd.Function):
    """rgb = RadianceNet(x, v, n, W8[1:] h7 + b8[1:]) on k_radiance_bf16 / k_radiance_bwd_bf16.  The weight inputs are the
    FOLDED matrices (autograd carries their gradients on to weight_g / weight_v); their gradients are plain GEMMs of
    the dumped deltas and activations."""

    @staticmethod
    def forward(ctx, model, x, v, n, h7, w8, b8, *rw_rb):
        from . import hip
        _, rad_blob = model.packed()
        x, v, n, h7 = x.contiguous(), v.contiguous(), n.detach().contiguous(), h7.detach().contiguous()
        rgb, dump = hip.radiance_fwd_dump(rad_blob, model.view_tiles, x, v, n, h7)
        ctx.model = model
        ctx.save_for_backward(x, v, n, h7, rgb, dump)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        from . import hip
        model = ctx.model
        x, v, n, h7, rgb, dump = ctx.saved_tensors
        M = x.shape[0]
        _, rad_blob = model.packed()
        g_rgb = g_rgb.contiguous()
        g_h7, g_n, bdump = hip.radiance_bwd(rad_blob, rgb, g_rgb, dump)
        perm = _unit_perm(x.device)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(256, device=x.device)
        nat = lambda m: m[:, inv]                                                   # unit order -> natural feature order
        acts = [nat(_dump_matrix(dump, s, M)) for s in range(5)]                     # f, r0, r1, r2, r3
        deltas = [nat(_dump_matrix(bdump, s, M)) for s in range(5)]                  # d3, d2, d1, d0, g_f
        d4 = g_rgb * rgb * (1.0 - rgb)
        rad = model.radiance_net
        ex = torch.cat([embed(x, rad.embed_multires), embed(v, rad.embed_multires_view), n], dim=-1)
        gw = [None] * 5
        gb = [None] * 5
        gw[4], gb[4] = d4.t() @ acts[4], d4.sum(0)
        gw[3], gb[3] = deltas[0].t() @ acts[3], deltas[0].sum(0)
        gw[2], gb[2] = deltas[1].t() @ acts[2], deltas[1].sum(0)
        gw[1], gb[1] = deltas[2].t() @ acts[1], deltas[2].sum(0)
        gw[0], gb[0] = torch.cat([deltas[3].t() @ ex, deltas[3].t() @ acts[0]], dim=1), deltas[3].sum(0)
        g_w8 = torch.cat([torch.zeros(1, 256, device=x.device), deltas[4].t() @ h7], dim=0)
        g_b8 = torch.cat([torch.zeros(1, device=x.device), deltas[4].sum(0)])
        out = [None, None, None, g_n, g_h7, g_w8, g_b8]
        for l in range(5):
            out += [gw[l], gb[l]]
        return tuple(out)


def radiance_forward_native(model, x, view_dirs, nabla, h7):
    surf_last = model.implicit_surface.surface_fc_layers[model.implicit_surface.D]
    w8 = torch._weight_norm(surf_last.weight_v, surf_last.weight_g, 0)
    args = []
    for lyr in model.radiance_net.layers:
        args += [torch._weight_norm(lyr.weight_v, lyr.weight_g, 0), lyr.bias]
    return RadianceNetFn.apply(model, x, view_dirs, nabla, h7, w8, surf_last.bias, *args)


def volsdf_point_forward(model, x, view_dirs, native_radiance: bool = False):
    """VolSDF.forward (volsdf.py:349-370): sphere clamp on sdf only, raw nabla into the radiance net."""
    sdf, nabla, feat = surface_forward_with_nablas(model.implicit_surface, x, return_h=native_radiance)
    d_bg = model.obj_bounding_radius - x.norm(dim=-1)
    sdf = torch.where(d_bg < sdf, d_bg, sdf)
    if native_radiance:
        return radiance_forward_native(model, x, view_dirs, nabla, feat), sdf, nabla
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
    if native_composite is None:
        native_composite = d_all.is_cuda
    native_radiance = bool(native_composite and getattr(model, "precision", "fp32") == "bf16x3")
    rad, sdf, nab = volsdf_point_forward(model, pts.reshape(-1, 3), v.reshape(-1, 3), native_radiance)
    rad, sdf, nab = rad.reshape(R, P, 3), sdf.reshape(R, P), nab.reshape(R, P, 3)
    alpha, beta = model.forward_ab()
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
