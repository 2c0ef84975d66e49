"""VolSDF renderer with the reference's boundary (models/frameworks/volsdf.py).

``volume_render(rays_o, rays_d, model, **kw) -> (rgb, depth, extras)`` takes the reference's keyword
set (:389-424) and returns the reference's ``extras`` keys (:566-594); ``SingleRenderer`` (:618-624) and
``get_model`` (:943-994) keep their shapes.  The whole chunk body (:448-596: sampling, network queries,
integration) is one call into the HIP library; Python only slices rays into chunks and reshapes.

Differences that cannot change results: launches hold up to 131,072 rays instead of
2,048-4,000 - the reference's ``rayschunk`` (render.py:488,614: 2048; ``val_rayschunk`` 1024, volsdf.py:990; 2000, :720) is a 24 GB
memory fit, rays are independent (:112) and the results here are chunk-invariant bit for bit (tests), so the argument is read as the
HINT it is: ``launch_rays`` (``honor_rayschunk=True`` slices exactly as asked; with ``perturb=True`` and no ``uniforms`` the chunking
decides which torch.rand call a ray's numbers come from, so the caller's value is kept there); the 256-d feature map the reference
materialises and drops in ``fine_sample`` never exists; weight_norm is folded once per weight update.
"""
from __future__ import annotations

import copy
from collections import OrderedDict

import torch
import torch.nn as nn

from . import hip
from .nets import VolSDF

DEFAULT_RAYSCHUNK = 131072      # a 480 x 270 frame in ONE set of sampler rounds (measured: 522.6 -> 519.7 ms against two 65,536-ray chunks; bit-identical)


def launch_rays(rayschunk, default: int, perturb: bool = False, uniforms=None, honor_rayschunk: bool = False) -> int:
    """Rays per library call for a caller's `rayschunk`.  The reference's values (1024 / 2048 / 2000: render.py:488,614, volsdf.py:720,990,
    train.py:189) size a chunk's activations for a 24 GB card and carry no semantics - every ray is rendered independently (volsdf.py:112,
    :599-610 only concatenates) - so a SMALLER value than this library's launch size is a hint that 288 GB of HBM does not need: the launch holds
    max(rayschunk, default) rays and the result is the same bit for bit (tests/test_gpu_configs.py).  Honoured exactly when the caller says
    so, and when the chunking is observable: perturb=True without `uniforms` draws one torch.rand block per chunk."""
    if not rayschunk:
        return int(default)
    r = int(rayschunk)
    if honor_rayschunk or (perturb and uniforms is None):
        return r
    return max(r, int(default))


def check_render_kwargs(use_nerfplusplus=False, use_view_dirs=True, **_):
    """The render arguments every VolSDF path of this package refuses (volume_render and the Trainer's staged pass 1 alike - ADVICE r05)."""
    if use_nerfplusplus:
        raise NotImplementedError("outside_scene: nerf++ is outside the hot-path scope (SURVEY.md 2, row 19)")
    if not use_view_dirs:
        raise NotImplementedError("use_view_dirs=False is not used by any reference config")


def volume_render(rays_o, rays_d, model: VolSDF, near=0.0, far=6.0, obj_bounding_radius=3.0, batched=False,
                  batched_info=None, require_nablas=False, calc_normal=True, use_view_dirs=True, rayschunk=None,
                  netchunk=1048576, white_bkgd=False, use_nerfplusplus=False, detailed_output=True,
                  show_progress=False, perturb=False, N_samples=128, N_importance=64, N_outside=32,
                  max_upsample_steps=5, max_bisection_steps=10, epsilon=0.1, k3_rays_chunk=8192, uniforms=None, honor_rayschunk=False,
                  **dummy_kwargs):
    """rays_o / rays_d: [(B,) N_rays, 3], rays_d un-normalised.  See module docstring.
    uniforms [N_rays, N_importance] (not a reference argument): with perturb=True, the uniform numbers of the final inverse-CDF
    samples, a row per ray, instead of a fresh torch.rand - how a test feeds the draws the reference made.
    rayschunk: the reference's memory hint - see launch_rays (honor_rayschunk=True: slice exactly as asked)."""
    check_render_kwargs(use_nerfplusplus=use_nerfplusplus, use_view_dirs=use_view_dirs)
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()) and rays_o.requires_grad:
        raise NotImplementedError("differentiable rays are not supported")
    lead = rays_o.shape[:-1]
    ro = rays_o.reshape(-1, 3).float().contiguous()
    rd = rays_d.reshape(-1, 3).float().contiguous()
    N = ro.shape[0]
    surf_blob, rad_blob = model.packed()
    radiance = model.packed_radiance()                     # None unless model.set_radiance_precision(...) chose another arithmetic for the radiance net
    sampler = model.packed_sampler()                       # None unless model.set_sampler_precision(...) chose another arithmetic for Algorithm 1
    alpha, beta = model.forward_ab()
    alpha, beta = float(alpha.detach()), float(beta.detach())
    chunk = launch_rays(rayschunk, DEFAULT_RAYSCHUNK, perturb, uniforms, honor_rayschunk)
    want_normal = bool(calc_normal and require_nablas)
    parts = []
    for i in range(0, N, chunk):
        # perturb (volsdf.py:122, rend_util.py:306-307): the 64 final samples invert the opacity CDF at uniform random numbers
        # instead of linspace(0, 1, 64); drawn here from torch's generator of the device, a row per ray
        if perturb and uniforms is not None:
            u_final = uniforms.reshape(N, N_importance)[i:i + chunk].to(device=ro.device, dtype=torch.float32).contiguous()
        else:
            u_final = torch.rand(min(chunk, N - i), N_importance, device=ro.device) if perturb else None
        # sampler: Algorithm 1 on its own blob / precision (model.set_sampler_precision; None = the model's): nerfart_volsdf_render_mixed_fwd
        parts.append(hip.volsdf_render(
            surf_blob, rad_blob, model.view_tiles, ro[i:i + chunk], rd[i:i + chunk], near=near, far=far,
            R_bg=obj_bounding_radius, alpha=alpha, beta=beta, eps=epsilon, n_samples=N_samples,
            n_importance=N_importance, max_upsample_steps=max_upsample_steps,
            max_bisection_steps=max_bisection_steps, white_bkgd=white_bkgd, calc_normal=want_normal,
            detailed=detailed_output, k3_rays_chunk=k3_rays_chunk, precision=model.precision_id, u_final=u_final, sampler=sampler,
            guard=model.sampler_guard if sampler is not None else 0.0, late_round=model.sampler_late_round if sampler is not None else 0,
            radiance=radiance, stats=model.render_stats))
    ret = OrderedDict()
    order = ["rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_surface", "implicit_nablas", "radiance",
             "alpha", "p_i", "visibility_weights", "d_vals", "sigma", "beta_map", "iter_usage"]
    for k in order:
        if k == "alpha":
            if detailed_output:
                ret["alpha"] = None
            continue
        if k in parts[0]:
            if k == "implicit_nablas" and not require_nablas:
                continue
            v = torch.cat([p[k] for p in parts], 0) if len(parts) > 1 else parts[0][k]
            ret[k] = v.reshape(*lead, *v.shape[1:])
    if detailed_output:
        ret["alpha"] = 1.0 - ret["p_i"]
        ret["beta_map"] = ret["beta_map"].unsqueeze(-1)          # reference shape [(B), N_rays, 1] (volsdf.py:590)
    return ret["rgb"], ret["depth_volume"], ret


class SingleRenderer(nn.Module):
    def __init__(self, model: VolSDF):
        super().__init__()
        self.model = model

    def forward(self, rays_o, rays_d, **kwargs):
        return volume_render(rays_o, rays_d, self.model, **kwargs)


def get_model(args, render_target=None):
    """(volsdf.py:943-994) args: the YAML config as an attribute dict (nerfart_amd/config.py).
    Returns (model, trainer, render_kwargs_train, render_kwargs_test, render_fn).  trainer = trainer.Trainer with
    `render_fn` set and, when `training.is_finetune`, the style losses configured from the YAML as the reference's
    Trainer.__init__ does (volsdf.py:638-645; criteria.build_style_loss - built at the first fine-tune step, so render.py with
    such a YAML loads no CLIP / VGG)."""
    m, t = args.model, args.training
    model_config = {
        "use_nerfplusplus": m.setdefault("outside_scene", "builtin") == "nerf++",
        "obj_bounding_radius": m.obj_bounding_radius,
        "W_geo_feat": m.setdefault("W_geometry_feature", 256),
        "speed_factor": t.setdefault("speed_factor", 1.0),
        "beta_init": t.setdefault("beta_init", 0.1),
    }
    s, r = m.surface, m.radiance
    model_config["surface_cfg"] = {
        "use_siren": s.setdefault("use_siren", m.setdefault("use_siren", False)),
        "embed_multires": s.setdefault("embed_multires", 6),
        "radius_init": s.setdefault("radius_init", 1.0),
        "geometric_init": s.setdefault("geometric_init", True),
        "D": s.setdefault("D", 8), "W": s.setdefault("W", 256), "skips": s.setdefault("skips", [4]),
    }
    model_config["radiance_cfg"] = {
        "use_siren": r.setdefault("use_siren", m.setdefault("use_siren", False)),
        "embed_multires": r.setdefault("embed_multires", -1),
        "embed_multires_view": r.setdefault("embed_multires_view", -1),
        "use_view_dirs": r.setdefault("use_view_dirs", True),
        "D": r.setdefault("D", 4), "W": r.setdefault("W", 256), "skips": r.setdefault("skips", []),
    }
    model = VolSDF(**model_config)
    # the arithmetic of the kernels (not a reference key; `model.precision` in the YAML or --model:precision overrides it): 'mixed' = the
    # shipped mode (nets._PackedModel.set_precision) - a model from get_model renders AND trains without a further call
    model.set_precision(m.get("precision", "mixed"))
    # `model.calibrate_sampler: true` (YAML of this repo, or `--model:calibrate_sampler true` on render.py's command line - the reference's own override syntax,
    # utils/io_util.py:270-319): the rendering form of `mixed` without touching render.py (nets.calibrate_sampler; the calibration itself runs at the first render,
    # on the weights loaded by then; a Trainer switches it back)
    if m.get("calibrate_sampler", False) and model.mode == "mixed":
        model.calibrate_sampler()
    render_kwargs_train = {
        "near": args.data.near, "far": args.data.far, "batched": True,
        "perturb": m.setdefault("perturb", True), "white_bkgd": m.setdefault("white_bkgd", False),
        "max_upsample_steps": m.setdefault("max_upsample_iter", 5),
        "use_nerfplusplus": model_config["use_nerfplusplus"], "obj_bounding_radius": m.obj_bounding_radius,
    }
    render_kwargs_test = copy.deepcopy(render_kwargs_train)
    render_kwargs_test["rayschunk"] = args.data.val_rayschunk
    render_kwargs_test["perturb"] = False
    renderer = SingleRenderer(model)
    from .trainer import Trainer
    trainer = Trainer(model)
    trainer.render_fn = renderer
    if bool(t.get("is_finetune", False)) and "finetune" in args:
        trainer.configure_style_loss(args, render_target if render_target is not None else [960, 540])     # volsdf.py:634-635
    return model, trainer, render_kwargs_train, render_kwargs_test, renderer
