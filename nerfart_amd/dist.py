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


def all_gather_tiles(t: torch.Tensor) -> torch.Tensor:
    """all_gather of equally shaped [n, C] tensors -> [world, n, C] ordered by rank: one all_gather_into_tensor (RCCL)."""
    w = world_size()
    if w == 1:
        return t[None]
    if _host_staged(t):
        h = t.detach().cpu().contiguous()
        out = torch.empty((w,) + tuple(h.shape), dtype=h.dtype)
        dist.all_gather_into_tensor(out.view(w * h.shape[0], *h.shape[1:]), h)
        return out.to(t.device)
    out = torch.empty((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out.view(w * t.shape[0], *t.shape[1:]), t.contiguous())
    return out


_KEY_WIDTHS = {"rgb": 3, "depth_volume": 1, "mask_volume": 1, "normals_volume": 3}      # columns of the [rays, C] tile
_PLANS = {}


class ShardPlan:
    """Everything about a (n_rays, tile, world) sharding that does not depend on the frame, built ONCE and kept on the device: this
    rank's ray indices, the size of the largest shard (the fixed all_gather size) and the permutation that turns the gathered
    [world * n_max, C] buffer back into ray order (one index_select per frame instead of `world` masked scatters)."""

    def __init__(self, n_rays: int, tile: int, rank_: int, world: int, device):
        per = [my_ray_indices(n_rays, tile, q, world) for q in range(world)]
        self.n_rays, self.tile, self.world, self.rank = n_rays, tile, world, rank_
        self.n_max = max(int(t.numel()) for t in per)
        self.idx = per[rank_].to(device)
        src = torch.empty(n_rays, dtype=torch.long)
        for q, t in enumerate(per):
            src[t] = q * self.n_max + torch.arange(t.numel())
        self.unshard = src.to(device)                      # full[i] = gathered.view(-1, C)[unshard[i]]


def shard_plan(n_rays: int, tile: int, device) -> ShardPlan:
    key = (n_rays, tile, rank(), world_size(), str(device))
    if key not in _PLANS:
        _PLANS[key] = ShardPlan(n_rays, tile, rank(), world_size(), device)
    return _PLANS[key]


def shard_rays(rays_o: torch.Tensor, rays_d: torch.Tensor, tile: int = 2048):
    """This rank's rays of a frame ([1, n_mine, 3] each) - call it when the frame's rays are made, outside the render step."""
    plan = shard_plan(rays_o.shape[-2], tile, rays_o.device)
    return rays_o[:, plan.idx].contiguous(), rays_d[:, plan.idx].contiguous()


def gather_frame(plan: ShardPlan, buf: torch.Tensor) -> torch.Tensor:
    """buf [n_max, C] (this rank's tile rows, zero padded) -> [n_rays, C] in ray order on every rank: ONE all_gather_into_tensor
    (RCCL over xGMI) + one index_select."""
    if plan.world == 1:
        return buf[: plan.n_rays]
    if _host_staged(buf):
        h = buf.detach().cpu().contiguous()
        out = torch.empty(plan.world * plan.n_max, buf.shape[1], dtype=buf.dtype)
        dist.all_gather_into_tensor(out, h)
        out = out.to(buf.device)
    else:
        out = torch.empty(plan.world * plan.n_max, buf.shape[1], dtype=buf.dtype, device=buf.device)
        dist.all_gather_into_tensor(out, buf.contiguous())
    return out.index_select(0, plan.unshard)


def render_sharded(render_fn, rays_o: torch.Tensor, rays_d: torch.Tensor, keys=("rgb", "depth_volume", "mask_volume", "normals_volume"),
                   tile: int = 2048, n_rays: int = None, **render_kwargs):
    """Ray-parallel render of one frame across all ranks.

    rays_o / rays_d: [1, N, 3], the same on every rank - or, with `n_rays` = N given, already this rank's shard of the frame
    (`shard_rays`: the gather then happens once where the rays are made, not inside the step).  Returns {key: [1, N, C]}
    assembled on every rank.  Ranks with fewer tiles pad to the largest shard so the all_gather is one fixed-size call; the
    sharding's index tensors are built once per (N, tile, world) (`shard_plan`)."""
    sharded_in = n_rays is not None
    N = n_rays if sharded_in else rays_o.shape[-2]
    dev = rays_o.device
    plan = shard_plan(N, tile, dev)
    n_mine = int(plan.idx.numel())
    if sharded_in:
        assert rays_o.shape[-2] == n_mine, (rays_o.shape, n_mine)
        ro, rd = rays_o, rays_d
    else:
        ro, rd = rays_o[:, plan.idx], rays_d[:, plan.idx]
    ex = render_fn(ro, rd, **render_kwargs)[2] if n_mine else {}
    widths = []
    for k in keys:
        if n_mine:
            v = ex[k][0]
            widths.append(1 if v.dim() == 1 else v.shape[1])
        else:
            widths.append(_KEY_WIDTHS[k])                   # a rank without rays still has to agree on the tile's columns
    C = sum(widths)
    buf = torch.zeros(plan.n_max, C, device=dev, dtype=torch.float32)
    c0 = 0
    for k, wd in zip(keys, widths):
        if n_mine:
            v = ex[k][0]
            buf[:n_mine, c0:c0 + wd] = v[:, None] if v.dim() == 1 else v
        c0 += wd
    full = gather_frame(plan, buf)
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
