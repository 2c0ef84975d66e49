"""Oracle: positional encoding + SDF / radiance MLPs on a reference-format state dict.

Test infrastructure only (see oracle/__init__.py).  Everything is a pure
function of ``sd`` = a dict of fp32 CPU tensors keyed exactly like the
reference checkpoint (``ln_beta`` | ``ln_s``,
``implicit_surface.surface_fc_layers.{i}.{bias,weight_g,weight_v}``,
``radiance_net.layers.{i}.{bias,weight_g,weight_v}``).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# a4  Embedder  (reference models/base.py:38-64, get_embedder :67-81)
# --------------------------------------------------------------------------
def embed(x: torch.Tensor, multires: int) -> torch.Tensor:
    """[..., C] -> [..., C*(1+2L)]: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...].

    multires < 0 is the identity (base.py:68-69).  Bands are the python floats
    ``2.**linspace(0, L-1, L)`` (base.py:39,44) i.e. exact powers of two.
    """
    if multires < 0:
        return x
    bands = (2.0 ** torch.linspace(0.0, multires - 1, multires)).numpy().tolist()
    parts = [x]
    for f in bands:
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, dim=-1)


def embed_dim(multires: int, c: int = 3) -> int:
    return c if multires < 0 else c * (1 + 2 * multires)


# --------------------------------------------------------------------------
# a5  weight-normed dense layer  (base.py:118-129, :226-227, :365-366)
# --------------------------------------------------------------------------
def folded_weight(sd: dict, prefix: str) -> torch.Tensor:
    """w[o,:] = g[o] * v[o,:] / ||v[o,:]||_2  (torch.nn.utils.weight_norm, dim=0)."""
    return torch._weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"], 0)


def n_layers(sd: dict, stem: str) -> int:
    i = 0
    while f"{stem}.{i}.bias" in sd:
        i += 1
    return i


def softplus100(x: torch.Tensor) -> torch.Tensor:
    """nn.Softplus(beta=100): x if 100x > 20 else log1p(exp(100x))/100 (base.py:202)."""
    return F.softplus(x, beta=100.0, threshold=20.0)


# --------------------------------------------------------------------------
# a6  ImplicitSurface.forward  (base.py:243-263)
# --------------------------------------------------------------------------
def surface_forward(sd: dict, x: torch.Tensor, multires: int = 6, skips=(4,),
                    stem: str = "implicit_surface.surface_fc_layers"):
    """x[..., 3] -> (sdf[...], feat[..., W_geo])."""
    nl = n_layers(sd, stem)          # D + 1 linear layers
    D = nl - 1
    e = embed(x, multires)
    h = e
    for i in range(D):
        if i in skips:
            # concat order [h, enc] and the divide by sqrt(2) are part of the contract (base.py:248-250)
            h = torch.cat([h, e], dim=-1) / np.sqrt(2)
        h = softplus100(F.linear(h, folded_weight(sd, f"{stem}.{i}"), sd[f"{stem}.{i}.bias"]))
    out = F.linear(h, folded_weight(sd, f"{stem}.{D}"), sd[f"{stem}.{D}.bias"])
    return out[..., 0], out[..., 1:]


# --------------------------------------------------------------------------
# a7  ImplicitSurface.forward_with_nablas  (base.py:265-282)
# --------------------------------------------------------------------------
def surface_forward_with_nablas(sd: dict, x: torch.Tensor, multires: int = 6, skips=(4,),
                                create_graph: bool = False):
    """x[..., 3] -> (sdf, nabla[..., 3] = d sdf / d x, feat).  Reverse-mode autograd like the reference."""
    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True) if not create_graph else x.requires_grad_(True)
        sdf, feat = surface_forward(sd, xg, multires, skips)
        nabla = torch.autograd.grad(sdf, xg, torch.ones_like(sdf), create_graph=create_graph,
                                    retain_graph=create_graph)[0]
    if not create_graph:
        sdf, nabla, feat = sdf.detach(), nabla.detach(), feat.detach()
    return sdf, nabla, feat


def surface_nablas_analytic(sd: dict, x: torch.Tensor, multires: int = 6, skips=(4,),
                            stem: str = "implicit_surface.surface_fc_layers"):
    """Forward-mode (tangent) evaluation of (sdf, nabla, feat) in fp64 - an independent
    cross-check of the autograd path and the algorithm the HIP kernel K3a uses."""
    xd = x.double()
    nl = n_layers(sd, stem)
    D = nl - 1
    L = multires
    parts, tparts = [xd], [torch.eye(3, dtype=torch.float64).expand(*xd.shape[:-1], 3, 3)]
    for k in range(L):
        f = 2.0 ** k
        s, c = torch.sin(xd * f), torch.cos(xd * f)
        parts += [s, c]
        tparts += [torch.diag_embed(f * c), torch.diag_embed(-f * s)]
    e = torch.cat(parts, -1)                      # [..., 39]
    te = torch.cat(tparts, -2)                    # [..., 39, 3]  d enc / d x
    h, th = e, te
    for i in range(D):
        if i in skips:
            h = torch.cat([h, e], -1) / math.sqrt(2)
            th = torch.cat([th, te], -2) / math.sqrt(2)
        w = folded_weight(sd, f"{stem}.{i}").double()
        z = h @ w.T + sd[f"{stem}.{i}.bias"].double()
        tz = torch.einsum("oi,...ij->...oj", w, th)
        h = F.softplus(z, beta=100.0, threshold=20.0)
        th = torch.sigmoid(100.0 * z)[..., None] * tz
    w = folded_weight(sd, f"{stem}.{D}").double()
    out = h @ w.T + sd[f"{stem}.{D}.bias"].double()
    tout = torch.einsum("oi,...ij->...oj", w, th)
    return out[..., 0], tout[..., 0, :], out[..., 1:]


# --------------------------------------------------------------------------
# a9  RadianceNet.forward  (base.py:372-391)
# --------------------------------------------------------------------------
def radiance_forward(sd: dict, x, view_dirs, normals, feat, multires: int = -1, multires_view: int = -1,
                     stem: str = "radiance_net.layers"):
    """cat[enc(x), enc_v(v), n, feat] -> D x (Linear+ReLU) -> Linear + Sigmoid -> rgb[..., 3]."""
    nl = n_layers(sd, stem)
    h = torch.cat([embed(x, multires), embed(view_dirs, multires_view), normals, feat], dim=-1)
    for i in range(nl):
        h = F.linear(h, folded_weight(sd, f"{stem}.{i}"), sd[f"{stem}.{i}.bias"])
        h = torch.sigmoid(h) if i == nl - 1 else torch.relu(h)
    return h


# --------------------------------------------------------------------------
# a8  VolSDF model level  (models/frameworks/volsdf.py:337-370)
# --------------------------------------------------------------------------
def volsdf_ab(sd: dict, speed_factor: float = 10.0):
    """(alpha, beta) = (1/beta, exp(ln_beta * speed))  (volsdf.py:337-339)."""
    beta = torch.exp(sd["ln_beta"] * speed_factor)
    return 1.0 / beta, beta


def volsdf_forward_surface(sd: dict, x, R: float = 3.0, multires: int = 6, skips=(4,)):
    """sdf = min(net(x), R - ||x||) - the built-in sphere background (volsdf.py:341-347)."""
    sdf, feat = surface_forward(sd, x, multires, skips)
    return torch.min(sdf, R - x.norm(dim=-1)), feat


def volsdf_forward(sd: dict, x, view_dirs, R: float = 3.0, multires: int = 6, skips=(4,),
                   rad_multires: int = -1, rad_multires_view: int = -1, create_graph: bool = False):
    """(radiance, sdf, nabla) with the sphere clamp applied to sdf only, raw nabla fed to the
    radiance net (volsdf.py:349-370).  create_graph=True keeps the autograd graph (through the nabla too:
    base.py:272-279 uses create_graph=True under grad mode) - the training rows (a19)."""
    sdf, nabla, feat = surface_forward_with_nablas(sd, x, multires, skips, create_graph=create_graph)
    d_bg = R - x.norm(dim=-1)
    sdf = torch.where(d_bg < sdf, d_bg, sdf)
    rad = radiance_forward(sd, x, view_dirs, nabla, feat, rad_multires, rad_multires_view)
    return rad, sdf, nabla


# --------------------------------------------------------------------------
# a18  NeuS model level  (models/frameworks/neus.py:111-123)
# --------------------------------------------------------------------------
def neus_s(sd: dict, speed_factor: float = 10.0):
    return torch.exp(sd["ln_s"] * speed_factor)


def neus_forward_radiance(sd: dict, x, view_dirs, multires: int = 6, skips=(4,),
                          rad_multires: int = -1, rad_multires_view: int = 4):
    _, nabla, feat = surface_forward_with_nablas(sd, x, multires, skips)
    return radiance_forward(sd, x, view_dirs, nabla, feat, rad_multires, rad_multires_view)
