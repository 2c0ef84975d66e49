"""Synthetic random-weight scenes for parity tests and the benchmark (no checkpoint ships with the
reference; SURVEY.md section 8d defines the scene):

  * model dims from the shipped YAMLs, ``torch.manual_seed(seed)``, the reference's geometric init;
  * SDF ``weight_v += 0.02 * std * N(0,1)`` (seed + 1) so the surface is not a perfect sphere;
  * ``ln_beta = ln(beta) / speed_factor`` with a trained-like beta (0.01 by default);
  * radiance ``weight_g *= 4`` so rgb has contrast (the untouched init renders ~0.5 grey);
  * pinhole camera looking at the origin from distance ``cam_dist`` (rend_util.look_at).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import rend_util
from .config import ConfigDict

VOLSDF_CFG = {
    "expname": "synthetic_volsdf", "device_ids": -1,
    "data": {"near": 0.0, "far": 6.0, "val_rayschunk": 1024, "batch_size": 1},
    "model": {"framework": "VolSDF", "obj_bounding_radius": 3.0, "outside_scene": "builtin", "max_upsample_iter": 6,
              "W_geometry_feature": 256,
              "surface": {"radius_init": 1.0, "D": 8, "skips": [4], "embed_multires": 6},
              "radiance": {"D": 4, "skips": [], "embed_multires": -1, "embed_multires_view": -1, "use_view_dirs": True}},
    "training": {"speed_factor": 10.0, "is_finetune": False},
}
NEUS_CFG = {
    "expname": "synthetic_neus", "device_ids": -1,
    "data": {"val_rayschunk": 512, "batch_size": 1},
    "model": {"framework": "NeuS", "obj_bounding_radius": 1.0, "variance_init": 0.05, "upsample_algo": "official_solution",
              "N_nograd_samples": 2048, "N_upsample_iters": 4,
              "surface": {"D": 8, "W": 256, "skips": [4], "radius_init": 0.5, "embed_multires": 6},
              "radiance": {"D": 4, "W": 256, "skips": [], "embed_multires": -1, "embed_multires_view": 4}},
    "training": {"speed_factor": 10.0, "with_mask": True, "is_finetune": False},
}


def synthetic_config(framework: str = "VolSDF") -> ConfigDict:
    return ConfigDict(VOLSDF_CFG if framework == "VolSDF" else NEUS_CFG)


def perturb_state(sd: dict, beta: float | None = 0.01, speed_factor: float = 10.0, seed: int = 1,
                  sdf_noise: float = 0.02, rad_gain: float = 4.0) -> dict:
    """Apply the scene's perturbation to a reference-format state dict (in a copy)."""
    g = torch.Generator().manual_seed(seed)
    out = {k: v.detach().clone() for k, v in sd.items()}
    for k in sorted(out):
        if k.startswith("implicit_surface.surface_fc_layers") and k.endswith("weight_v"):
            v = out[k]
            out[k] = v + sdf_noise * v.std() * torch.randn(v.shape, generator=g)
        if k.startswith("radiance_net.layers") and k.endswith("weight_g"):
            out[k] = out[k] * rad_gain
    if beta is not None and "ln_beta" in out:
        out["ln_beta"] = torch.tensor([math.log(beta) / speed_factor], dtype=torch.float32)
    return out


def build_model(framework: str = "VolSDF", seed: int = 0, beta: float | None = 0.01, device=None, precision: str = "fp32"):
    """(model, render_kwargs_test, render_fn) with the synthetic scene's weights.  precision: 'fp32' (the default HERE: what the exact-parity
    tests ask for), 'bf16x3', 'mixed' (the product's default, frameworks.get_model), 'fp16x2' - nets._PackedModel.set_precision."""
    from .frameworks import get_model
    cfg = synthetic_config(framework)
    torch.manual_seed(seed)
    model, _, _, rk_test, render_fn = get_model(cfg)
    model.load_state_dict(perturb_state(model.state_dict(), beta=beta, seed=seed + 1))
    if device is not None:
        model.to(device)
    model.set_precision(precision)
    return model, rk_test, render_fn


def camera(H: int, W: int, cam_dist: float = 2.5, focal_scale: float = 1.25, angle: float = 0.0):
    """c2w [4,4], K [4,4] float32: pinhole fx = fy = focal_scale * W, principal point at the centre."""
    loc = np.array([cam_dist * math.sin(angle), 0.0, -cam_dist * math.cos(angle)])
    c2w = torch.from_numpy(rend_util.look_at(loc, np.zeros(3))).float()
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = focal_scale * W
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    return c2w, K


def spiral(n_views: int, cam_dist: float = 2.5):
    """n_views poses on a circle of radius cam_dist around the origin, all looking at it (cfg 5)."""
    return [2.0 * math.pi * i / n_views for i in range(n_views)]
