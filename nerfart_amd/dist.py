"""Multi-GPU: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The reference scales with single-process ``nn.DataParallel`` over the ray axis (volsdf.py:632-633) or
image-parallel DDP (train.py:83-87,154-155).  Rays are independent (fine_sample is per ray, volsdf.py:112),
so here a frame is cut into tiles of ``tile`` consecutive rays dealt round-robin to the ranks (work per ray
varies 1.0-4.2 GFLOP with the up-sampling rounds and is spatially coherent - contiguous 1/N splits would
imbalance), every rank renders its tiles with the single-GPU path, and ONE all_gather of [rays, C] fp32
tiles (3.6 MB per 480x270 frame in total) reassembles the frame on every rank.  No collective sits inside
the data path of a tile.  Training adds one flat all-reduce of the 796,347 gradients (3.2 MB).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def rank() -> int:
    return dist.get_rank() if is_initialized() else 0


def world_size() -> int:
    return dist.get_world_size() if is_initialized() else 1


def init(backend: str | None = None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    if is_initialized() or int(os.environ.get("WORLD_SIZE", 1)) <= 1:
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group(backend)


def tile_assignment(n_rays: int, tile: int, world: int):
    """[(start, stop, owner)] for consecutive ray tiles dealt round-robin."""
    return [(s, min(s + tile, n_rays), (i % world)) for i, s in enumerate(range(0, n_rays, tile))]


def my_ray_indices(n_rays: int, tile: int, rank_: int, world: int, device=None) -> torch.Tensor:
    parts = [torch.arange(s, e, device=device) for s, e, o in tile_assignment(n_rays, tile, world) if o == rank_]
    return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.long, device=device)


def _host_staged(t: torch.Tensor) -> bool:
    """gloo has no all_gather for device tensors: stage through the host (tests that run several ranks on ONE GPU; the
    production backend is RCCL, which moves device memory directly)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_gather_tiles(t: torch.Tensor):
    """all_gather of equally shaped [n, C] tensors -> list ordered by rank (one RCCL call)."""
    if world_size() == 1:
        return [t]
    if _host_staged(t):
        h = t.detach().cpu().contiguous()
        out = [torch.empty_like(h) for _ in range(world_size())]
        dist.all_gather(out, h)
        return [o.to(t.device) for o in out]
    out = [torch.empty_like(t) for _ in range(world_size())]
    dist.all_gather(out, t.contiguous())
    return out


def render_sharded(render_fn, rays_o: torch.Tensor, rays_d: torch.Tensor, keys=("rgb", "depth_volume", "mask_volume", "normals_volume"),
                   tile: int = 2048, **render_kwargs):
    """Ray-parallel render of one frame across all ranks.

    rays_o / rays_d: [1, N, 3] (same on every rank).  Returns {key: [1, N, C]} assembled on every rank.
    Ranks with fewer tiles pad to the largest shard so the all_gather is one fixed-size call."""
    w, r = world_size(), rank()
    N = rays_o.shape[-2]
    dev = rays_o.device
    idx = my_ray_indices(N, tile, r, w, dev)
    _, _, ex = render_fn(rays_o[:, idx], rays_d[:, idx], **render_kwargs) if idx.numel() else (None, None, {})
    cols, widths = [], []
    for k in keys:
        if idx.numel():
            v = ex[k][0]
            v = v[:, None] if v.dim() == 1 else v
        else:
            v = None
        cols.append(v)
        widths.append(None if v is None else v.shape[1])
    if w == 1:
        return {k: (c[None] if c.shape[1] > 1 else c[None, :, 0]) for k, c in zip(keys, cols)}
    # widths must agree across ranks even if a rank had no rays: exchange them
    wt = torch.tensor([x if x is not None else 0 for x in widths], dtype=torch.long)
    wt = wt if dist.get_backend() == "gloo" else wt.to(dev)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    widths = wt.tolist()
    n_max = max(len(my_ray_indices(N, tile, q, w)) for q in range(w))
    C = sum(widths)
    buf = torch.zeros(n_max, C, device=dev, dtype=torch.float32)
    if idx.numel():
        buf[: idx.numel()] = torch.cat(cols, dim=1)
    gathered = all_gather_tiles(buf)
    full = torch.empty(N, C, device=dev, dtype=torch.float32)
    for q in range(w):
        qi = my_ray_indices(N, tile, q, w, dev)
        full[qi] = gathered[q][: qi.numel()]
    out, c0 = {}, 0
    for k, wd in zip(keys, widths):
        v = full[:, c0:c0 + wd]
        out[k] = v[None] if wd > 1 else v[None, :, 0]
        c0 += wd
    return out


def allreduce_gradients(params, average: bool = False):
    """One flat all-reduce of all gradients (a single 3.2 MB bucket: latency-bound on xGMI, so never split)."""
    if world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    if _host_staged(flat):
        h = flat.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        flat = h.to(flat.device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= world_size()
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()
