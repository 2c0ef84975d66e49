"""Differentiable evaluation of the per-sample networks and the compositing - the part of the fine-tune step
that needs parameter gradients (reference Trainer.forward pass 2, models/frameworks/volsdf.py:753-770,
neus.py:520-576).

Two formulations live here:
* the NATIVE pass 2 (`volsdf_backward_samples_native` / `neus_backward_samples_native`, `surface_param_backward`,
  `GradAccumulator`): thin wrappers over the RAY-LEVEL C entry points nerfart_volsdf_render_bwd / nerfart_neus_render_bwd /
  nerfart_sdf_param_bwd (csrc/render_backward.hip: the whole kernel sequence - points, radiance forward with dumps, compositor
  backward, radiance backward, cotangents, second-order SDF sweeps, weight-gradient reductions - behind one call per launch
  group) and nerfart_fold_weight_grads / nerfart_weight_norm_bwd; what `Trainer` runs on split-bf16 models;
* the AUTOGRAD formulation (`surface_forward*`, `radiance_forward`, `volsdf_render_samples`, `neus_render_samples`):
  PyTorch autograd over rocBLAS GEMMs, including the double backward through the SDF net that the eikonal term and
  the normal input of the radiance net need.  It follows the reference formulas line by line: the cross-check every
  native piece is tested against at full size, and the path of the fp32-exact precision (`Trainer(native=False)`).
"""
import numpy as np
import torch
import torch.nn.functional as F


_FREQS = {}


def _freqs(device, multires: int) -> torch.Tensor:
    key = (str(device), multires)
    if key not in _FREQS:
        _FREQS[key] = (2.0 ** torch.arange(multires, dtype=torch.float32))[:, None].to(device)      # [K, 1]
    return _FREQS[key]


def embed(x: torch.Tensor, multires: int) -> torch.Tensor:
    """models/base.py:46-64: cat[x, sin(2^k x), cos(2^k x), ...], k = 0..multires-1; identity for multires = -1.
    (All octaves in one sin and one cos launch; 2^k x is exact, so this is the loop's result bit for bit.)"""
    if multires < 0:
        return x
    xf = x[..., None, :] * _freqs(x.device, multires).to(x.dtype)                  # [..., K, 3]
    return torch.cat([x, torch.stack([torch.sin(xf), torch.cos(xf)], dim=-2).flatten(-3)], dim=-1)


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


# ---- native pass 2: thin wrappers over the ray-level C entry points (csrc/render_backward.hip) ----------------------------
class GradAccumulator:
    """The RAW parameter-gradient buffer of a step (include/nerfart_hip.h: nerfart_pass2_raw_layout).  The ray-level backward
    entry points accumulate into it launch group after launch group; everything downstream - un-permuting the kernels' unit
    order, the weight_norm chain rule, the alpha / beta (or s) chain rule - is linear and runs ONCE in flush()
    (nerfart_fold_weight_grads + nerfart_weight_norm_bwd per layer)."""

    def __init__(self):
        self.raw = None

    def buffer(self, device) -> torch.Tensor:
        from . import hip
        if self.raw is None:
            self.raw = hip.new_raw(device)
        return self.raw

    def _scalar(self, which: int) -> torch.Tensor:
        from . import hip
        return self.raw[hip.raw_layout()[0]["scalars"] + which]

    def flush(self, model):
        """.grad += of the model's parameters; empties the accumulator."""
        from . import hip
        raw, self.raw = self.raw, None
        if raw is None:
            return
        surf, rad = model.implicit_surface, model.radiance_net
        folded, offs = hip.fold_weight_grads(raw, surf.embed_multires, rad.embed_multires_view)
        layers = list(surf.surface_fc_layers) + list(rad.layers)
        train_rad = any(p.requires_grad for p in rad.parameters())
        # nerfart_fold_weight_grads is built for the shipped architecture (D 8, skip at 4, multires 6; radiance x-embedding -1): a model of
        # another shape must not get silently misaligned gradients
        if rad.embed_multires != -1 or len(offs) < 2 * len(layers) + 1:
            raise RuntimeError("GradAccumulator.flush: the gradient fold is built for radiance embed_multires = -1 and the 9 + 5 layer nets")
        for k, lyr in enumerate(layers):
            if lyr.weight_v.numel() != offs[2 * k + 1] - offs[2 * k] or lyr.bias.numel() != offs[2 * k + 2] - offs[2 * k + 1]:
                raise RuntimeError(f"GradAccumulator.flush: layer {k} is {tuple(lyr.weight_v.shape)}, the folded layout holds "
                                   f"{offs[2 * k + 1] - offs[2 * k]} weights / {offs[2 * k + 2] - offs[2 * k + 1]} biases")
        for k, lyr in enumerate(layers):
            if k >= len(surf.surface_fc_layers) and not train_rad:
                break
            out_f, in_f = lyr.weight_v.shape
            dW = folded[offs[2 * k]: offs[2 * k] + out_f * in_f].view(out_f, in_f)
            g_v, g_g = hip.weight_norm_bwd(dW, lyr.weight_v.detach().contiguous(), lyr.weight_g.detach().contiguous())
            _add_grad(lyr.weight_v, g_v)
            _add_grad(lyr.weight_g, g_g)
            _add_grad(lyr.bias, folded[offs[2 * k + 1]: offs[2 * k + 1] + out_f].clone())
        sc = raw[hip.raw_layout()[0]["scalars"]:]
        if hasattr(model, "ln_beta"):
            a, b = model.forward_ab()                                  # alpha = 1 / beta, beta = exp(ln_beta speed) (volsdf.py:337-339)
            torch.autograd.backward([a, b], [sc[hip.RAW_G_ALPHA].reshape(a.shape), sc[hip.RAW_G_BETA].reshape(b.shape)])
        else:
            s_t = model.forward_s()                                    # s = exp(ln_s speed) (neus.py:111-112)
            torch.autograd.backward([s_t], [sc[hip.RAW_G_S].reshape(s_t.shape)])


def _add_grad(p, g):
    if p.requires_grad:
        p.grad = g.reshape(p.shape) if p.grad is None else p.grad + g.reshape(p.shape)


def _need_split_bf16(model, who: str):
    """The native backward entry points take no precision argument and read split-bf16 blobs only."""
    if getattr(model, "precision", None) != "bf16x3":
        raise RuntimeError(f"{who}: native pass 2 needs model.set_precision('bf16x3') (model is at {getattr(model, 'precision', None)!r})")


def _eikonal_terms(nab, R: int, P: int, w_eikonal: float, group_rays):
    """(sum over patches of w * mean_patch((|nabla| - 1)^2), its gradient w.r.t. the nablas [R P, 3]) when the R rays of
    one launch are `group_rays`-ray patches of the reference's pass 2 (each patch has its OWN mean; the last may be
    ragged).  group_rays None: one patch.  (The torch statement of what nerfart_volsdf_pass2_cotangents computes: tests.)"""
    nn_ = nab.norm(dim=-1)
    err = nn_ - 1.0
    if group_rays is None or group_rays >= R:
        return w_eikonal * (err ** 2).mean(), (w_eikonal * 2.0 / nn_.numel()) * (err / nn_)[:, None] * nab
    wr = torch.full((R,), 1.0 / (group_rays * P), device=nab.device)
    tail = R % group_rays
    if tail:
        wr[R - tail:] = 1.0 / (tail * P)
    eik = w_eikonal * ((err ** 2).reshape(R, P).sum(1) * wr).sum()
    coef = (2.0 * w_eikonal) * wr[:, None].expand(R, P).reshape(-1)
    return eik, (coef * err / nn_)[:, None] * nab


def volsdf_backward_samples_native(model, rays_o, rays_d, d_all, g_rgb, w_eikonal=0.1, use_eikonal=True, white_bkgd=False, ab=None,
                                   nbar_extra=None, accum=None, state=None, eik_group_rays=None, g_acc=None):
    """Pass 2 of the fine-tune step for one launch group of rays: ONE call of nerfart_volsdf_render_bwd, which accumulates into the
    step's raw buffer what rgb.backward(g_rgb) and (w_eikonal * MSE(|nabla|, 1)).backward() accumulate (volsdf.py:759-770).
    rays_d: un-normalised (the entry point normalises, as the forward does).  Returns the eikonal loss (a 0-d tensor: no host
    synchronisation in here; `ab` = (alpha, beta) as Python floats if the caller already has them).
    nbar_extra [R, P, 3]: a further cotangent of the nablas (the reconstruction branch's one-sample-per-ray eikonal term).
    accum: a GradAccumulator shared by the launch groups of a step - the caller flushes it once; None: .grad is updated here.
    state: (sdf, nablas, h7) of these points kept from pass 1 (same weights: identical values) - skips their re-evaluation.
    eik_group_rays: the rays are several `eik_group_rays`-ray patches of the reference in one launch (the eikonal mean is
    per patch); the returned loss is then the SUM over those patches."""
    from . import hip
    _need_split_bf16(model, "volsdf_backward_samples_native")
    surf_blob, rad_blob = model.packed()
    if ab is None:
        alpha, beta = model.forward_ab()
        ab = (float(alpha), float(beta))
    acc = accum if accum is not None else GradAccumulator()
    raw = acc.buffer(d_all.device)
    before = acc._scalar(hip.RAW_EIKONAL).clone()
    hip.volsdf_render_bwd(surf_blob, rad_blob, model.view_tiles, model.implicit_surface.embed_multires, rays_o.contiguous(), rays_d.contiguous(),
                          d_all.contiguous(), g_rgb.contiguous(), raw, R_bg=model.obj_bounding_radius, alpha=ab[0], beta=ab[1],
                          white_bkgd=white_bkgd, w_eikonal=w_eikonal if use_eikonal else 0.0, eik_group_rays=eik_group_rays or 0,
                          train_radiance=any(p.requires_grad for p in model.radiance_net.parameters()),
                          g_acc=None if g_acc is None else g_acc.contiguous(),
                          g_n_extra=None if nbar_extra is None else nbar_extra.reshape(-1, 3).contiguous(),
                          state=None if state is None else tuple(t.contiguous() for t in state))
    eik = acc._scalar(hip.RAW_EIKONAL) - before
    if accum is None:
        acc.flush(model)
    return eik


def neus_backward_samples_native(model, rays_o, rays_d, d_all, g_rgb, w_eikonal=0.1, use_eikonal=True, white_bkgd=False, s_val=None, accum=None,
                                 eik_group_rays=None, g_acc=None, state=None):
    """Pass 2 for one launch group of NeuS rays: ONE call of nerfart_neus_render_bwd (neus.py:310-395, :520-576): SDF + nablas at
    the P samples (alpha, eikonal), SDF + nablas + radiance at the P-1 mid-points.  Radiance-net gradients are produced only if its
    parameters require grad (the fine-tune step freezes it, neus.py:455-456; reconstruction trains it).  g_acc [R]: a cotangent of
    the opacity mask_volume (the mask BCE of the reconstruction objective).  state: (sdf, nablas[, ...]) at the samples kept from
    pass 1.  Returns the eikonal loss (0-d)."""
    from . import hip
    _need_split_bf16(model, "neus_backward_samples_native")
    surf_blob, rad_blob = model.packed()
    if s_val is None:
        s_val = float(model.forward_s())
    acc = accum if accum is not None else GradAccumulator()
    raw = acc.buffer(d_all.device)
    before = acc._scalar(hip.RAW_EIKONAL).clone()
    hip.neus_render_bwd(surf_blob, rad_blob, model.view_tiles, model.implicit_surface.embed_multires, rays_o.contiguous(), rays_d.contiguous(),
                        d_all.contiguous(), g_rgb.contiguous(), raw, s=s_val, white_bkgd=white_bkgd, w_eikonal=w_eikonal if use_eikonal else 0.0,
                        eik_group_rays=eik_group_rays or 0, train_radiance=any(p.requires_grad for p in model.radiance_net.parameters()),
                        g_acc=None if g_acc is None else g_acc.contiguous(),
                        state=None if state is None else (state[0].contiguous(), state[1].contiguous()))
    eik = acc._scalar(hip.RAW_EIKONAL) - before
    if accum is None:
        acc.flush(model)
    return eik


def surface_param_backward(model, pts, nbar, sbar=None, hbar7=None, accum=None):
    """Parameter gradients of sbar . sdf + hbar7 . h7 + nbar . grad_x sdf at free points (nerfart_sdf_param_bwd): the uniformly
    drawn eikonal points of the reconstruction objective (volsdf.py:799-806)."""
    from . import hip
    _need_split_bf16(model, "surface_param_backward")
    acc = accum if accum is not None else GradAccumulator()
    surf_blob, _ = model.packed()
    step = 1 << 21
    for i in range(0, pts.shape[0], step):
        sl = slice(i, i + step)
        hip.sdf_param_bwd(surf_blob, model.implicit_surface.embed_multires, pts[sl].contiguous(), nbar[sl].contiguous(), acc.buffer(pts.device),
                          sbar=None if sbar is None else sbar[sl].contiguous(), hbar7=None if hbar7 is None else hbar7[sl].contiguous())
    if accum is None:
        acc.flush(model)


class RadianceNetFn(torch.autograd.Function):
    """rgb = RadianceNet(x, v, n, W8[1:] h7 + b8[1:]) on k_radiance_bf16 / k_radiance_bwd_bf16 (nerfart_radiance_param_bwd).  The
    weight inputs are the FOLDED matrices (autograd carries their gradients on to weight_g / weight_v); their gradients come
    from the raw buffer through nerfart_fold_weight_grads."""

    @staticmethod
    def forward(ctx, model, x, v, n, h7, w8, b8, *rw_rb):
        from . import hip
        _, rad_blob = model.packed()
        x, v, n, h7 = x.contiguous(), v.contiguous(), n.detach().contiguous(), h7.detach().contiguous()
        rgb = hip.radiance_fwd(rad_blob, model.view_tiles, x, v, n, h7, precision=model.precision_id)
        ctx.model = model
        ctx.save_for_backward(x, v, n, h7)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        from . import hip
        model = ctx.model
        x, v, n, h7 = ctx.saved_tensors
        _, rad_blob = model.packed()
        raw = hip.new_raw(x.device)
        _, g_h7, g_n = hip.radiance_param_bwd(rad_blob, model.view_tiles, x, v, n, h7, g_rgb.contiguous(), raw)
        surf, rad = model.implicit_surface, model.radiance_net
        folded, offs = hip.fold_weight_grads(raw, surf.embed_multires, rad.embed_multires_view)

        def piece(k, lyr):
            out_f, in_f = lyr.weight_v.shape
            return folded[offs[2 * k]: offs[2 * k] + out_f * in_f].view(out_f, in_f), folded[offs[2 * k + 1]: offs[2 * k + 1] + out_f]

        g_w8, g_b8 = piece(surf.D, surf.surface_fc_layers[surf.D])      # row 0 (the sdf row) is zero: no SDF sweep ran
        out = [None, None, None, g_n, g_h7, g_w8, g_b8]
        for l, lyr in enumerate(rad.layers):
            out += list(piece(surf.D + 1 + l, lyr))
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
