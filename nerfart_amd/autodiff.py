"""Differentiable evaluation of the per-sample networks and the compositing - the part of the fine-tune step
that needs parameter gradients (reference Trainer.forward pass 2, models/frameworks/volsdf.py:753-770,
neus.py:520-576).

Two formulations live here:
* the NATIVE pass 2 (`volsdf_backward_samples_native` / `neus_backward_samples_native`, `GradAccumulator`,
  `*_weight_grads_raw / _finish`): hand-written backward kernels (hip.radiance_fwd_dump / radiance_bwd / sdf_fwd2 /
  sdf_bwd2 / *_composite_bwd) + the hand-written weight-gradient reduction (nerfart_wgrad_bf16) over their point-major dumps, read
  in place; what `Trainer` runs
  on split-bf16 models;
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


# ---- radiance net on the hand-written kernels (forward with activation dumps, backward chain, GEMM operands) ----
_PERMS = {}        # device -> (perm, inv) on that device: a host -> device copy per call would synchronise the stream


def _perms(device):
    key = str(device)
    if key not in _PERMS:
        from .packing import unit_feature_hidden
        perm = torch.tensor([unit_feature_hidden(u, g, e) for u in range(8) for g in range(4) for e in range(8)])
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(256)
        _PERMS[key] = (perm.to(device), inv.to(device))
    return _PERMS[key]


def _embed_width(multires: int) -> int:
    return 3 if multires < 0 else 3 + 6 * multires


def _wgrad_narrow(Z: torch.Tensor, A64: torch.Tensor, c: int, split: bool, n_mats: int = 1, z_stride: int = 0, want_cs: bool = False, cs_rows: int = 0):
    """Z_m^T [A narrow] -> [n_mats, 256, c] (+ column sums) through the hand-written kernel, hi + lo columns of A summed."""
    from . import hip
    rows = A64.shape[0]
    w, cs = hip.wgrad(Z, A64, n_mats, rows, 64, z_stride, 0, cs_rows=cs_rows, want_cs=want_cs)
    w = w[:, :, :c] + w[:, :, 32:32 + c] if split else w[:, :, :c]
    return w, cs


def _inv_perm(device):
    return _perms(device)[1]


def _unperm(w: torch.Tensor, inv: torch.Tensor, rows: bool = True, cols: bool = True) -> torch.Tensor:
    """A small GEMM result indexed by unit-order features -> natural feature order."""
    if rows:
        w = w[inv]
    if cols:
        w = w[:, inv]
    return w


def radiance_weight_grads_raw(model, x, v, n, h7, rgb, g_rgb, dump, bdump, a7_units=None):
    """Reductions of the deltas (k_radiance_bwd_bf16's dump) against the activations (k_radiance_bf16<dump>'s), read in place;
    the big operands stay in the kernels' unit order, the narrow ones come from csrc/pass2_operands.hip in one pass each.
    a7_units [>= M rows, 256] bf16: the layer-7 activation in unit order if the caller has it (k_sdf_fwd2_bf16 dumps it: slot 7,
    value rows) - otherwise h7 is permuted and narrowed here.  Returns the raw fp32 results - LINEAR in the cotangents, so the
    results of several patches may be summed (GradAccumulator) before radiance_weight_grads_finish."""
    from . import hip
    M = x.shape[0]
    acts = _dump_all(dump)                                                  # f, r0, r1, r2, r3   [5, M_pad, 256]
    deltas = _dump_all(bdump)                                               # d0, d1, d2, d3, g_f (0 for the padded points)
    Mp = acts.shape[1]
    rad = model.radiance_net
    slot = Mp * 512                                                         # bytes between the dumps' slots
    # (d0, f), (d1, r0), (d2, r1), (d3, r2) and the column sums of d0..d3 in one pass over the dumps (csrc/wgrad.hip)
    ww, cs03 = hip.wgrad(deltas[0], acts[0], 4, Mp, 256, slot, slot, cs_rows=Mp, want_cs=True)
    if a7_units is None:
        a7_units = torch.zeros(Mp, 256, dtype=torch.bfloat16, device=x.device)
        a7_units[:M] = h7[:, _perms(x.device)[0]].to(torch.bfloat16)
    rows7 = min(a7_units.shape[0], Mp)                                      # both cover the M points; rows beyond them are zero
    wh7, cs4 = hip.wgrad(deltas[4], a7_units, 1, rows7, 256, 0, 0, cs_rows=rows7, want_cs=True)
    cs = torch.cat([cs03, cs4], dim=0)                                      # [5, 256]
    D4, b4, _ = hip.wgrad_operand_rgb_delta(rgb, g_rgb, Mp)                 # g_rgb rgb (1 - rgb): the output sigmoid; its column sums
    w4 = _wgrad_narrow(acts[4], D4, 3, True)[0][0].t()                      # [3, 256] = d4^T r3
    nex = _embed_width(rad.embed_multires) + _embed_width(rad.embed_multires_view) + 3
    EX = hip.wgrad_operand_inputs(x, rad.embed_multires, v, rad.embed_multires_view, n, Mp)
    wex = _wgrad_narrow(deltas[0], EX, nex, nex <= 32)[0][0]                # [256, nex] = d0^T [x | v | n]
    return [ww, cs, w4, b4, wex, wh7[0]]


def radiance_weight_grads_finish(model, raw):
    """Gradients of the FOLDED radiance weights / biases and of rows 1.. of the last SDF layer from the raw GEMM results:
    the 256-wide axes are re-indexed from unit order to the reference's feature order."""
    ww, cs, w4, b4, wex, wh7 = raw
    inv = _inv_perm(ww.device)
    gw, gb = [None] * 5, [None] * 5
    gw[4], gb[4] = _unperm(w4, inv, rows=False), b4
    for l in (1, 2, 3):
        gw[l], gb[l] = _unperm(ww[l], inv), cs[l][inv]
    gw[0] = torch.cat([_unperm(wex, inv, cols=False), _unperm(ww[0], inv)], dim=1)
    gb[0] = cs[0][inv]
    g_w8 = torch.cat([torch.zeros(1, 256, device=ww.device), _unperm(wh7, inv)], dim=0)
    g_b8 = torch.cat([torch.zeros(1, device=ww.device), cs[4][inv]])
    return gw, gb, g_w8, g_b8


def radiance_weight_grads(model, x, v, n, h7, rgb, g_rgb, dump, bdump):
    return radiance_weight_grads_finish(model, radiance_weight_grads_raw(model, x.contiguous(), v.contiguous(), n.contiguous(), h7, rgb.contiguous(),
                                                                          g_rgb.contiguous(), dump, bdump))


_F2_SLOTS, _R2_SLOTS = 16, 8


def _pair_all(dump: torch.Tensor, slots: int, nslots: int) -> torch.Tensor:
    """[nslots, 2 Mp, 256] bf16 VIEW of a column-pair kernel's dump (Mp = M rounded up to the kernels' 64-point tiles; the
    padded points carry zero tangents and zero cotangents, so their rows add nothing to any GEMM): the kernels store
    point-major rows, the first Mp of a slot from one column of the pair and the next Mp from the other (k_sdf_fwd2:
    [a; adot], k_sdf_bwd2: [zbar; t d]); features in unit order."""
    return dump.view(torch.bfloat16).view(slots, -1, 256)[:nslots]


def _dump_all(dump: torch.Tensor) -> torch.Tensor:
    """[5, M_pad, 256] bf16 VIEW of the radiance kernels' point-major dumps (unit order)."""
    return dump.view(torch.bfloat16).view(5, -1, 256)


def surface_weight_grads_raw(model, pts, sbar, hbar7, nbar, return_a7: bool = False):
    """k_sdf_fwd2_bf16 / k_sdf_bwd2_bf16 + the reductions over their dumps for the cotangents (sbar of sdf [M], hbar7 of the
    layer-7 activation [M,256], nbar of grad_x sdf [M,3]).  dW_l = zbar_l^T a_{l-1} + (t_l d_l)^T adot_{l-1} is ONE reduction
    over the 2 Mp stacked rows of the two dumps, read in place; the big operands stay in unit order.  Returns the raw fp32
    results - linear in the cotangents, so patches may be summed before surface_weight_grads_finish; with return_a7 also the
    layer-7 activation [Mp, 256] bf16 in unit order (a view of the forward dump) for radiance_weight_grads_raw."""
    from . import hip
    surf = model.implicit_surface
    surf_blob, _ = model.packed()
    M = pts.shape[0]
    pts, nbar, sbar = pts.contiguous(), nbar.contiguous(), sbar.contiguous()
    f2 = hip.sdf_fwd2(surf_blob, pts, nbar)
    r2 = hip.sdf_bwd2(surf_blob, hbar7.contiguous(), sbar, f2)
    Mp = (M + 63) // 64 * 64                                            # the kernels' tiles; padded rows are zero where it matters
    RZ = _pair_all(r2, _R2_SLOTS, 8)                                    # [8, 2 Mp, 256]: 65535 * [zbar_l; t_l d_l]
    FA = _pair_all(f2, _F2_SLOTS, 8)                                    # [8, 2 Mp, 256]: [a_l; adot_l]
    slot = 2 * Mp * 512                                                 # bytes between the dumps' slots
    # layers 1..7 against the previous layer's (a | adot), with the column sums of zbar_1..7 (first Mp rows) in the same pass
    ww, cs17 = hip.wgrad(RZ[1], FA[0], 7, 2 * Mp, 256, slot, slot, cs_rows=Mp, want_cs=True)
    # layers 0 and 4 (four slots apart) against the encoding [e; edot], one shared narrow operand; column sums of zbar_0
    nenc = _embed_width(surf.embed_multires)
    E2 = hip.wgrad_operand_embed_pair(pts, nbar, Mp, surf.embed_multires)
    we, cs04 = _wgrad_narrow(RZ[0], E2, nenc, nenc <= 32, n_mats=2, z_stride=4 * slot, want_cs=True, cs_rows=Mp)
    cs = torch.cat([cs04[0:1], cs17], dim=0)                            # [8, 256]: sum_p zbar_l
    # w8 row: a7^T sbar + column sums of adot7 = FA[7]^T [sbar; 1]
    w8 = _wgrad_narrow(FA[7], hip.wgrad_operand_sbar_ones(sbar, Mp), 1, True)[0][0][:, 0]
    raw = [ww, we, cs, w8, sbar.sum()]
    return (raw, FA[7][:Mp]) if return_a7 else raw


def surface_weight_grads_finish(model, raw):
    """(dW[0..8], db[0..8]) of the FOLDED SDF-net weights / biases from the raw results; layer 8 holds the sdf row only
    (rows 1.. belong to radiance_weight_grads)."""
    ww, we, cs, w8row, b8sum = raw
    surf = model.implicit_surface
    inv = _inv_perm(ww.device)
    rs2 = 1.0 / np.sqrt(2.0)
    sc = 1.0 / 65535.0
    dW, db = [None] * 9, [None] * 9
    for l in range(8):
        out_dim = surf.surface_fc_layers[l].out_features
        if l == 0:
            w = _unperm(we[0], inv, cols=False)
        elif l in surf.skips:
            hw = surf.W - we.shape[2]
            w = torch.cat([_unperm(ww[l - 1], inv)[:, :hw], _unperm(we[1], inv, cols=False)], dim=1) * rs2
        else:
            w = _unperm(ww[l - 1], inv)
        dW[l] = (w * sc)[:out_dim]
        db[l] = (cs[l][inv] * sc)[:out_dim]
    w8 = torch.zeros(surf.surface_fc_layers[8].out_features, 256, device=ww.device)
    w8[0] = w8row[inv]
    b8 = torch.zeros(w8.shape[0], device=ww.device)
    b8[0] = b8sum
    dW[8], db[8] = w8, b8
    return dW, db


def surface_weight_grads(model, pts, sbar, hbar7, nbar):
    return surface_weight_grads_finish(model, surface_weight_grads_raw(model, pts, sbar, hbar7, nbar))


class GradAccumulator:
    """Sums the raw weight-gradient results of pass 2 over the patches of a step.  Everything downstream of them - un-permuting
    the 256-wide axes, the weight_norm chain rule, the alpha / beta (or s) chain rule - is linear and runs ONCE in flush()
    instead of once per patch (~150 tiny launches per patch otherwise)."""

    def __init__(self):
        self.sums = {}

    def add(self, key: str, raw):
        raw = [t if isinstance(t, torch.Tensor) else torch.as_tensor(t) for t in raw]
        if key not in self.sums:
            self.sums[key] = [t.clone() for t in raw]
        else:
            torch._foreach_add_(self.sums[key], raw)

    def flush(self, model):
        """.grad += of the model's parameters; empties the accumulator."""
        if "surf" in self.sums:
            dW, db = surface_weight_grads_finish(model, self.sums["surf"])
            if "rad" in self.sums:
                gw, gb, g_w8, g_b8 = radiance_weight_grads_finish(model, self.sums["rad"])
                dW[8] = dW[8] + g_w8                              # the geometry-feature rows of the last SDF layer
                db[8] = db[8] + g_b8
                if any(p.requires_grad for p in model.radiance_net.parameters()):
                    accumulate_folded_grads(list(model.radiance_net.layers), gw, gb)
            accumulate_folded_grads(list(model.implicit_surface.surface_fc_layers), dW, db)
        if "ab" in self.sums:
            g_ab = self.sums["ab"][0]
            a, b = model.forward_ab()
            torch.autograd.backward([a, b], [g_ab[0:1].reshape(a.shape), g_ab[1:2].reshape(b.shape)])
        if "s" in self.sums:
            s_t = model.forward_s()
            torch.autograd.backward([s_t], [self.sums["s"][0].reshape(s_t.shape)])
        self.sums = {}


def accumulate_folded_grads(layers, dW, db):
    """p.grad += for weight_g / weight_v / bias of weight-normed layers, given gradients of the folded weights."""
    folded = [torch._weight_norm(l.weight_v, l.weight_g, 0) for l in layers]
    torch.autograd.backward(folded, [g.to(f.dtype) for g, f in zip(dW, folded)])
    for l, g in zip(layers, db):
        if l.bias.requires_grad:
            l.bias.grad = g.clone() if l.bias.grad is None else l.bias.grad + g


def _eikonal_terms(nab, R: int, P: int, w_eikonal: float, group_rays):
    """(sum over patches of w * mean_patch((|nabla| - 1)^2), its gradient w.r.t. the nablas [R P, 3]) when the R rays of
    one launch are `group_rays`-ray patches of the reference's pass 2 (each patch has its OWN mean; the last may be
    ragged).  group_rays None: one patch."""
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


def _need_split_bf16(model, who: str):
    """The native backward entry points take no precision argument and read split-bf16 blobs only."""
    if getattr(model, "precision", None) != "bf16x3":
        raise RuntimeError(f"{who}: native pass 2 needs model.set_precision('bf16x3') (model is at {getattr(model, 'precision', None)!r})")


def volsdf_backward_samples_native(model, rays_o, rays_dn, d_all, g_rgb, w_eikonal=0.1, use_eikonal=True, white_bkgd=False, ab=None,
                                   nbar_extra=None, accum=None, state=None, eik_group_rays=None):
    """Pass 2 of the fine-tune step for one patch, entirely on the hand-written kernels + GEMMs: accumulates into .grad what
    rgb.backward(g_rgb) and (w_eikonal * MSE(|nabla|, 1)).backward() accumulate (volsdf.py:759-770).  Returns the eikonal loss
    (a 0-d tensor: no host synchronisation in here; `ab` = (alpha, beta) as Python floats if the caller already has them).
    nbar_extra [R, P, 3]: a further cotangent of the nablas (the reconstruction branch's one-sample-per-ray eikonal term).
    accum: a GradAccumulator shared by the patches of a step - the caller flushes it once; None: .grad is updated here.
    state: (sdf, nablas, h7) of these points kept from pass 1 (same weights: identical values) - skips their re-evaluation.
    eik_group_rays: the rays are several `eik_group_rays`-ray patches of the reference in one launch (the eikonal mean is
    per patch); the returned loss is then the SUM over those patches."""
    from . import hip
    _need_split_bf16(model, "volsdf_backward_samples_native")
    R, P = d_all.shape
    d_all = d_all.contiguous()
    pts, v = hip.ray_points(rays_o.contiguous(), rays_dn.contiguous(), d_all)
    surf_blob, rad_blob = model.packed()
    Rbg = model.obj_bounding_radius
    with torch.no_grad():
        sdf, nab, h7 = state if state is not None else hip.sdf_nabla_fwd(surf_blob, pts, Rbg, precision=model.precision_id)
        rgb_pt, dump = hip.radiance_fwd_dump(rad_blob, model.view_tiles, pts, v, nab, h7)
        if ab is None:
            alpha, beta = model.forward_ab()
            ab = (float(alpha), float(beta))
        g_sdf, g_rad, g_ab = hip.volsdf_composite_bwd(d_all, sdf.reshape(R, P), rgb_pt.reshape(R, P, 3), ab[0], ab[1],
                                                      g_rgb.contiguous(), white_bkgd)
        g_rad = g_rad.reshape(-1, 3)
        g_h7, g_n, bdump = hip.radiance_bwd(rad_blob, rgb_pt, g_rad, dump)
        # sbar: no gradient to the net where sdf = min(net, R - |x|) took the sphere; nbar = g_n + the eikonal term's gradient
        sbar, nbar, eik_ray = hip.volsdf_pass2_cotangents(pts, sdf.reshape(-1), g_sdf.reshape(-1), nab, g_n, R, P, Rbg,
                                                          w_eikonal if use_eikonal else 0.0, eik_group_rays,
                                                          g_n_extra=None if nbar_extra is None else nbar_extra.reshape(-1, 3).contiguous())
        eik = eik_ray.sum()
        acc = accum if accum is not None else GradAccumulator()
        surf_raw, a7 = surface_weight_grads_raw(model, pts, sbar, g_h7, nbar, return_a7=True)
        acc.add("surf", surf_raw)
        acc.add("rad", radiance_weight_grads_raw(model, pts, v, nab, h7, rgb_pt, g_rad, dump, bdump, a7_units=a7))
        acc.add("ab", [g_ab])
    if accum is None:
        acc.flush(model)
    return eik


def neus_backward_samples_native(model, rays_o, rays_dn, d_all, g_rgb, w_eikonal=0.1, use_eikonal=True, white_bkgd=False, s_val=None, accum=None,
                                 eik_group_rays=None, g_acc=None, state=None):
    """Pass 2 for one NeuS patch on the hand-written kernels + GEMMs (neus.py:310-395, :520-576): SDF + nablas at the P
    samples (alpha, eikonal), SDF + nablas + radiance at the P-1 mid-points.  Radiance-net gradients are accumulated only if
    its parameters require grad (the fine-tune step freezes it, neus.py:455-456; reconstruction trains it).  g_acc [R]: a
    cotangent of the opacity mask_volume (the mask BCE of the reconstruction objective).  Returns the eikonal loss (0-d)."""
    from . import hip
    _need_split_bf16(model, "neus_backward_samples_native")
    R, P = d_all.shape
    rays_o, rays_dn = rays_o.contiguous(), rays_dn.contiguous()
    pts, _ = hip.ray_points(rays_o, rays_dn, d_all.contiguous(), want_view=False)
    pts_m, v_m = hip.ray_points(rays_o, rays_dn, (0.5 * (d_all[..., 1:] + d_all[..., :-1])).contiguous())
    surf_blob, rad_blob = model.packed()
    with torch.no_grad():
        if s_val is None:
            s_val = float(model.forward_s())
        # state: (sdf, nablas, _) at the samples kept from pass 1 (same weights: identical values)
        sdf, nab, _ = state if state is not None else hip.sdf_nabla_fwd(surf_blob, pts, 0.0, want_h7=False, precision=model.precision_id)
        _, nab_m, h7_m = hip.sdf_nabla_fwd(surf_blob, pts_m, 0.0, precision=model.precision_id)
        rgb_m, dump = hip.radiance_fwd_dump(rad_blob, model.view_tiles, pts_m, v_m, nab_m, h7_m)
        g_sdf, g_rad, g_s = hip.neus_composite_bwd(sdf.reshape(R, P), rgb_m.reshape(R, P - 1, 3), s_val, g_rgb.contiguous(), white_bkgd,
                                                      g_acc=None if g_acc is None else g_acc.contiguous())
        g_h7, g_n, bdump = hip.radiance_bwd(rad_blob, rgb_m, g_rad.reshape(-1, 3), dump)
        nbar = torch.zeros_like(nab)
        eik = torch.zeros((), device=pts.device)
        if use_eikonal:
            eik, nbar = _eikonal_terms(nab, R, P, w_eikonal, eik_group_rays)
        # samples: cotangents of sdf (alpha) and of the nablas (eikonal); mid-points: of h7 and of the normal (radiance)
        acc = accum if accum is not None else GradAccumulator()
        acc.add("surf", surface_weight_grads_raw(model, pts, g_sdf.reshape(-1), torch.zeros(pts.shape[0], 256, device=pts.device), nbar))
        surf_raw, a7 = surface_weight_grads_raw(model, pts_m, torch.zeros(pts_m.shape[0], device=pts.device), g_h7, g_n, return_a7=True)
        acc.add("surf", surf_raw)
        acc.add("rad", radiance_weight_grads_raw(model, pts_m, v_m, nab_m, h7_m, rgb_m, g_rad.reshape(-1, 3).contiguous(), dump, bdump, a7_units=a7))
        acc.add("s", [g_s])
    if accum is None:
        acc.flush(model)
    return eik


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
        _, rad_blob = model.packed()
        g_rgb = g_rgb.contiguous()
        g_h7, g_n, bdump = hip.radiance_bwd(rad_blob, rgb, g_rgb, dump)
        gw, gb, g_w8, g_b8 = radiance_weight_grads(model, x, v, n, h7, rgb, g_rgb, dump, bdump)
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
