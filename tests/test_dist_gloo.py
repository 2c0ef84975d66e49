"""world_size-2 gloo tests (CPU) of the ray-sharding layer: tile assignment, padded all_gather reassembly,
flat gradient all-reduce.  The per-tile renderer is replaced by a deterministic per-ray function - the
sharding logic, not the HIP kernels, is what runs here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_render(rays_o, rays_d, **kw):
    # any per-ray function: results must not depend on which other rays share the call
    rgb = torch.sin(rays_d * 3.0 + rays_o)
    depth = (rays_d ** 2).sum(-1)
    ex = {"rgb": rgb, "depth_volume": depth, "mask_volume": depth * 0.5, "normals_volume": torch.cos(rays_d)}
    return rgb, depth, ex


def _worker(rank, world, port, n_rays, tile, q):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from nerfart_amd import dist as nd
    nd.init("gloo")
    g = torch.Generator().manual_seed(0)
    o = torch.randn(1, n_rays, 3, generator=g)
    d = torch.randn(1, n_rays, 3, generator=g)
    out = nd.render_sharded(_fake_render, o, d, tile=tile)
    _, _, ref = _fake_render(o, d)
    ok = all(torch.equal(out[k], ref[k]) for k in ("rgb", "depth_volume", "mask_volume", "normals_volume"))
    # the same frame with the rays sharded where they are made (outside the step): identical result; the plan is cached
    om, dm = nd.shard_rays(o, d, tile=tile)
    out2 = nd.render_sharded(_fake_render, om, dm, tile=tile, n_rays=n_rays)
    ok = ok and all(torch.equal(out2[k], ref[k]) for k in ("rgb", "depth_volume", "mask_volume", "normals_volume"))
    plan = nd.shard_plan(n_rays, tile, o.device)
    ok = ok and plan is nd.shard_plan(n_rays, tile, o.device) and om.shape[1] == plan.idx.numel()
    ok = ok and torch.equal(torch.sort(torch.cat([nd.my_ray_indices(n_rays, tile, q, world) for q in range(world)])).values, torch.arange(n_rays))
    g3 = nd.all_gather_tiles(torch.full((3, 2), float(rank)))
    ok = ok and g3.shape == (world, 3, 2) and all(float(g3[q].mean()) == q for q in range(world))
    # gradient all-reduce
    p = [torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(2, 3))]
    for i, t in enumerate(p):
        t.grad = torch.full_like(t, float(rank + 1 + i))
    nd.allreduce_gradients(p)
    ok = ok and torch.equal(p[0].grad, torch.full((5,), 3.0)) and torch.equal(p[1].grad, torch.full((2, 3), 5.0))
    q.put((rank, bool(ok), int(nd.my_ray_indices(n_rays, tile, rank, world).numel())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rays,tile", [(1000, 64), (129, 128), (64, 2048)])
def test_ray_sharded_render_world2(n_rays, tile):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rays, tile, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) == n_rays


def test_tile_assignment_is_a_partition():
    from nerfart_amd import dist as nd
    for n, tile, w in ((129600, 2048, 8), (518400, 2048, 8), (100, 7, 3)):
        idx = torch.cat([nd.my_ray_indices(n, tile, r, w) for r in range(w)])
        assert torch.equal(torch.sort(idx).values, torch.arange(n))
        sizes = [nd.my_ray_indices(n, tile, r, w).numel() for r in range(w)]
        assert max(sizes) - min(sizes) <= tile
