"""Ray utilities with the reference's call shapes (utils/rend_util.py), computed by the HIP library.

get_rays (:112-165) and lin2img (:238-248) are the two the hot path uses; ``look_at`` (:44-53) is the
host-side camera helper the synthetic benchmark scene and render paths are built with.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip


def normalize(vec):
    return vec / (np.linalg.norm(vec, axis=-1, keepdims=True) + 1e-9)


def look_at(cam_location, point, up=np.array([0., -1., 0.])):
    """OpenCV-convention camera-to-world matrix looking from cam_location at point (rend_util.py:30-53)."""
    fwd = normalize(point - cam_location)
    rx = normalize(np.cross(up, fwd))
    ry = normalize(np.cross(fwd, rx))
    m = np.stack((rx, ry, fwd, cam_location), axis=-1)
    return np.concatenate((m, np.array([[0., 0., 0., 1.]])), axis=-2)


def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """(rend_util.py:76-93) - host-side 3x3 from a quaternion pose; a handful of flops per camera."""
    q = torch.nn.functional.normalize(q, dim=-1)
    qr, qi, qj, qk = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = torch.ones(*q.shape[:-1], 3, 3, dtype=q.dtype, device=q.device)
    R[..., 0, 0] = 1 - 2 * (qj ** 2 + qk ** 2); R[..., 0, 1] = 2 * (qj * qi - qk * qr); R[..., 0, 2] = 2 * (qi * qk + qr * qj)
    R[..., 1, 0] = 2 * (qj * qi + qk * qr); R[..., 1, 1] = 1 - 2 * (qi ** 2 + qk ** 2); R[..., 1, 2] = 2 * (qj * qk - qi * qr)
    R[..., 2, 0] = 2 * (qk * qi - qj * qr); R[..., 2, 1] = 2 * (qj * qk + qi * qr); R[..., 2, 2] = 1 - 2 * (qi ** 2 + qj ** 2)
    return R


def get_rays(c2w: torch.Tensor, intrinsics: torch.Tensor, H: int, W: int, N_rays: int = -1):
    """c2w [B,4,4] or [B,7], intrinsics [B,4,4] (on the GPU) -> rays_o, rays_d [B,N,3], select_inds [B,N].

    N_rays > 0 draws a random pixel subset with independent randint on rows and columns, shared by
    the batch, exactly as the reference (rend_util.py:137-140)."""
    dev = c2w.device
    if c2w.shape[-1] == 7:
        p = torch.eye(4, device=dev).repeat(*c2w.shape[:-1], 1, 1).float()
        p[..., :3, :3] = quat_to_rot(c2w[..., :4])
        p[..., :3, 3] = c2w[..., 4:]
    else:
        p = c2w
    prefix = p.shape[:-2]
    pb = p.reshape(-1, 4, 4).float().contiguous()
    Kb = intrinsics.reshape(-1, 4, 4).float().contiguous().to(dev)
    if N_rays > 0:
        N_rays = min(N_rays, H * W)
        hs = torch.randint(0, H, size=[N_rays]).to(dev)
        ws = torch.randint(0, W, size=[N_rays]).to(dev)
        sel = (hs * W + ws).contiguous()
    else:
        sel = None
    os_, ds_ = [], []
    for b in range(pb.shape[0]):
        o, d = hip.get_rays(pb[b], Kb[b], H, W, sel)
        os_.append(o); ds_.append(d)
    n = os_[0].shape[0]
    rays_o = torch.stack(os_).reshape(*prefix, n, 3)
    rays_d = torch.stack(ds_).reshape(*prefix, n, 3)
    inds = (sel if sel is not None else torch.arange(H * W, device=dev)).expand(*prefix, n)
    return rays_o, rays_d, inds


def lin2img(tensor: torch.Tensor, H: int, W: int, batched=False, B=None):
    """[(B,) H*W, C] -> [(B,) C, H, W]  (rend_util.py:238-248) - pure layout."""
    *_, num_samples, channels = tensor.shape
    assert num_samples == H * W
    if batched:
        if B is None:
            B = tensor.shape[0]
        else:
            tensor = tensor.view([B, num_samples // B, channels])
        return tensor.permute(0, 2, 1).view([B, channels, H, W])
    return tensor.permute(1, 0).view([channels, H, W])
