"""NeuS renderer with the reference's boundary (models/frameworks/neus.py:142-432), on the HIP library.

``volume_render(rays_o, rays_d, model, **kw) -> (rgb, depth, extras)``; ``extras`` carries the reference's
keys (:384-407).  upsample_algo: 'official_solution' (the one the shipped configs use, :275-303), 'direct_use' (:242-255), 'direct_more'
(:259-269); N_outside = 0 (configs: ``outside_scene`` absent, ``with_mask: True``).
"""
from __future__ import annotations

import copy
from collections import OrderedDict

import torch
import torch.nn as nn

from . import hip
from .nets import NeuS

DEFAULT_RAYSCHUNK = 65536


def volume_render(rays_o, rays_d, model: NeuS, obj_bounding_radius=1.0, batched=False, batched_info=None,
                  calc_normal=False, use_view_dirs=True, rayschunk=None, netchunk=1048576, white_bkgd=False,
                  near_bypass=None, far_bypass=None, detailed_output=True, show_progress=False, perturb=False,
                  fixed_s_recp=1 / 64., N_samples=64, N_importance=64, N_outside=0, upsample_algo="official_solution",
                  N_nograd_samples=2048, N_upsample_iters=4, k3_rays_chunk=8192, uniforms=None, honor_rayschunk=False, **dummy_kwargs):
    """uniforms [N_rays, N_importance] (not a reference argument): with perturb=True, the uniform numbers of the up-sampling rounds
    (round k reads columns k * N_importance / N_upsample_iters ...), a row per ray, instead of a fresh torch.rand.
    rayschunk (512 in the NeuS YAMLs' val_rayschunk, neus.py:747): the reference's memory hint - volsdf.launch_rays."""
    if upsample_algo not in hip.NEUS_UPSAMPLE_ALGOS:
        raise NotImplementedError(f"upsample_algo {upsample_algo!r} (neus.py:303: the reference raises too)")
    if N_outside > 0 or near_bypass is not None or far_bypass is not None:
        raise NotImplementedError("NeuS render: N_outside=0, no near/far bypass (the reference configs) are on the HIP path")
    if not use_view_dirs:
        raise NotImplementedError("use_view_dirs=False is not used by any reference config")
    lead = rays_o.shape[:-1]
    ro = rays_o.reshape(-1, 3).float().contiguous()
    rd = rays_d.reshape(-1, 3).float().contiguous()
    N = ro.shape[0]
    surf_blob, rad_blob = model.packed()
    if model.packed_sampler() is not None:
        raise NotImplementedError("set_sampler_precision is a VolSDF measurement variant (Algorithm 1); NeuS's up-sampling runs at the model's precision")
    s = float(model.forward_s().detach())
    from .volsdf import launch_rays
    chunk = launch_rays(rayschunk, DEFAULT_RAYSCHUNK, perturb, uniforms, honor_rayschunk)
    parts = []
    for i in range(0, N, chunk):
        # perturb (neus.py:296, rend_util.py:269-272): every up-sampling round inverts its CDF at uniform random numbers
        if perturb and uniforms is not None:
            u_new = uniforms.reshape(N, N_importance)[i:i + chunk].to(device=ro.device, dtype=torch.float32).contiguous()
        else:
            u_new = torch.rand(min(chunk, N - i), N_importance, device=ro.device) if perturb else None
        parts.append(hip.neus_render(
            surf_blob, rad_blob, model.view_tiles, ro[i:i + chunk], rd[i:i + chunk],
            obj_bounding_radius=obj_bounding_radius, s=s, n_samples=N_samples, n_importance=N_importance,
            n_upsample_iters=N_upsample_iters, white_bkgd=white_bkgd, calc_normal=calc_normal,
            detailed=detailed_output, k3_rays_chunk=k3_rays_chunk, precision=model.precision_id, u_new=u_new,
            upsample_algo=upsample_algo, n_nograd_samples=N_nograd_samples, fixed_s_recp=fixed_s_recp))
    ret = OrderedDict()
    for k in ["rgb", "depth_volume", "mask_volume", "normals_volume", "implicit_nablas", "implicit_surface", "radiance",
              "alpha", "cdf", "visibility_weights", "d_final", "d_all"]:      # d_all: the P sample depths (an extra key)
        if k in parts[0]:
            v = torch.cat([p[k] for p in parts], 0) if len(parts) > 1 else parts[0][k]
            ret[k] = v.reshape(*lead, *v.shape[1:])
    return ret["rgb"], ret["depth_volume"], ret


class SingleRenderer(nn.Module):
    def __init__(self, model: NeuS):
        super().__init__()
        self.model = model

    def forward(self, rays_o, rays_d, **kwargs):
        return volume_render(rays_o, rays_d, self.model, **kwargs)


def get_model(args, render_target=None):
    """(neus.py:693-750)"""
    m, t = args.model, args.training
    model_config = {
        "obj_bounding_radius": m.obj_bounding_radius,
        "W_geo_feat": m.setdefault("W_geometry_feature", 256),
        "use_outside_nerf": False,
        "speed_factor": t.setdefault("speed_factor", 1.0),
        "variance_init": m.setdefault("variance_init", 0.05),
    }
    s, r = m.surface, m.radiance
    model_config["surface_cfg"] = {
        "embed_multires": s.setdefault("embed_multires", 6), "radius_init": s.setdefault("radius_init", 1.0),
        "geometric_init": s.setdefault("geometric_init", True), "D": s.setdefault("D", 8), "W": s.setdefault("W", 256),
        "skips": s.setdefault("skips", [4]),
    }
    model_config["radiance_cfg"] = {
        "embed_multires": r.setdefault("embed_multires", -1),
        "embed_multires_view": r.setdefault("embed_multires_view", -1),
        "use_view_dirs": r.setdefault("use_view_dirs", True),
        "D": r.setdefault("D", 4), "W": r.setdefault("W", 256), "skips": r.setdefault("skips", []),
    }
    model = NeuS(**model_config)
    # the arithmetic of the kernels (not a reference key; `model.precision` in the YAML or --model:precision overrides it): 'mixed' = the
    # shipped mode (nets._PackedModel.set_precision) - a model from get_model renders AND trains without a further call
    model.set_precision(m.get("precision", "mixed"))
    render_kwargs_train = {
        "upsample_algo": m.setdefault("upsample_algo", "official_solution"),
        "N_nograd_samples": m.setdefault("N_nograd_samples", 2048),
        "N_upsample_iters": m.setdefault("N_upsample_iters", 4),
        "N_outside": m.setdefault("N_outside", 0),
        "obj_bounding_radius": args.data.setdefault("obj_bounding_radius", 1.0),     # (sic) data, not model: neus.py:738
        "batched": args.data.batch_size is not None,
        "perturb": m.setdefault("perturb", True), "white_bkgd": m.setdefault("white_bkgd", False),
    }
    render_kwargs_test = copy.deepcopy(render_kwargs_train)
    render_kwargs_test["rayschunk"] = args.data.val_rayschunk
    render_kwargs_test["perturb"] = False
    from .trainer import Trainer
    renderer = SingleRenderer(model)
    # neus.py:455-456: the radiance net is frozen when fine-tuning, trained otherwise
    trainer = Trainer(model, freeze_radiance=bool(t.get("is_finetune", False)))
    trainer.render_fn = renderer
    if bool(t.get("is_finetune", False)) and "finetune" in args:
        trainer.configure_style_loss(args, render_target if render_target is not None else [960, 540])     # neus.py:432-446
    return model, trainer, render_kwargs_train, render_kwargs_test, renderer
