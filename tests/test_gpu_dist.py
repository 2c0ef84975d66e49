"""The ray-sharded render and fine-tune step with two ranks on ONE GPU (gloo for the collectives, staged through the host; the
production backend is RCCL with one GPU per rank): the gradients after the all-reduce equal the single-process step's."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir):
    import numpy as np
    import torch.distributed as dist
    from nerfart_amd import scene, rend_util, dist as nd
    from nerfart_amd.trainer import Trainer
    try:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        dev = "cuda"
        model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision="bf16x3")
        H, W = 10, 7
        c2w, K = scene.camera(H, W)
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        target = (torch.rand(1, H * W, 3, generator=torch.Generator().manual_seed(2)) * 0.2 + 0.6).to(dev)
        loss_fn = lambda pred, gt: ((pred - gt) ** 2).mean()
        kw = {k: v for k, v in rk.items() if k != "rayschunk"}
        keys = ("rgb", "depth_volume", "mask_volume", "normals_volume")
        frame = nd.render_sharded(render_fn, o, d, keys=keys, tile=16, detailed_output=False, require_nablas=True, calc_normal=True, **kw)
        tr = Trainer(model, pass2_rays=8, patches_per_launch=2)
        model.zero_grad()
        out = tr.finetune_step(render_fn, o, d, target, H, loss_fn, **kw)       # 70 rays: 8-ray patches dealt 5 (38 rays) + 4 (32)
        assert nd.world_size() == 2 and len(nd.my_ray_indices(H * W, 8, rank, world)) in (38, 32)
        sharded = {n: p.grad.clone() for n, p in model.named_parameters()}
        rgb_sharded = out["rgb"].clone()
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            with torch.no_grad():
                _, _, ex = render_fn(o, d, detailed_output=False, require_nablas=True, calc_normal=True, **kw)
            for k in keys:                                            # the sharded frame is the single-process frame, ray for ray
                np.testing.assert_array_equal(frame[k].cpu().numpy(), ex[k].cpu().numpy(), err_msg=k)
            model.zero_grad()
            ref = Trainer(model, pass2_rays=8, patches_per_launch=2).finetune_step(render_fn, o, d, target, H, loss_fn, **kw)
            np.testing.assert_allclose(rgb_sharded.cpu().numpy(), ref["rgb"].cpu().numpy(), atol=1e-6)
            assert abs(out["loss"] - ref["loss"]) < 1e-7
            for n, p in model.named_parameters():
                rel = float((sharded[n] - p.grad).norm() / (p.grad.norm() + 1e-12))
                # ranks own whole patches of the single-process patch grid (tile = pass2_rays): same per-patch eikonal means,
                # same per-point operands - only the summation order of the weight-gradient GEMMs / the all-reduce differs
                assert rel < 1e-4, (n, rel)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    except Exception as e:                                            # surfaced by the parent
        import traceback
        open(os.path.join(out_dir, f"err{rank}"), "w").write(traceback.format_exc())
        raise


def test_sharded_finetune_step_two_ranks_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.get_context("spawn")
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    errs = [open(tmp_path / f).read() for f in os.listdir(tmp_path) if f.startswith("err")]
    assert not errs, errs[0]
    assert sorted(f for f in os.listdir(tmp_path)) == ["ok0", "ok1"]


def _perturb_worker(rank, world, port, out_dir):
    """The sharded fine-tune step at the reference's DEFAULT render_kwargs_train (perturb=True): pass 2 back-propagates through its own random
    samples, taken from pass 1's run of Algorithm 1 (Trainer.render_two_draws) on the rank's rays.  Fed per-ray uniform tables (indexed by
    the GLOBAL ray id), the all-reduced gradients equal the single-process step's."""
    import torch.distributed as dist
    from nerfart_amd import scene, rend_util, dist as nd
    from nerfart_amd.trainer import Trainer
    try:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        dev = "cuda"
        model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision="mixed")
        H, W = 10, 7
        c2w, K = scene.camera(H, W)
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        g = torch.Generator().manual_seed(2)
        target = (torch.rand(1, H * W, 3, generator=g) * 0.2 + 0.6).to(dev)
        tables = {1: torch.rand(H * W, 64, generator=g), 2: torch.rand(H * W, 64, generator=g)}
        loss_fn = lambda pred, gt: ((pred - gt) ** 2).mean()
        kw = dict({k: v for k, v in rk.items() if k != "rayschunk"}, perturb=True)
        tr = Trainer(model, pass2_rays=8, patches_per_launch=2)
        assert tr.shares_algorithm1(kw)
        # the SAME frame-relative source as the single-process step below (round 6, ADVICE r05: the sharded step maps its tiles' rays onto the frame's rows
        # itself - Trainer._global_rays - instead of handing the source shard-relative offsets)
        tr.uniform_source = lambda p, first, count, n, dv: tables[p][first:first + count, :n].to(dv)
        model.zero_grad()
        out = tr.finetune_step(render_fn, o, d, target, H, loss_fn, **kw)
        sharded = {n: p.grad.clone() for n, p in model.named_parameters()}
        rgb_sharded, loss_sharded = out["rgb"].clone(), out["loss"]
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            ref_tr = Trainer(model, pass2_rays=8, patches_per_launch=2)
            ref_tr.uniform_source = lambda p, first, count, n, dv: tables[p][first:first + count, :n].to(dv)
            model.zero_grad()
            ref = ref_tr.finetune_step(render_fn, o, d, target, H, loss_fn, **kw)
            assert float((rgb_sharded - ref["rgb"]).abs().max()) <= 1e-6 and abs(loss_sharded - ref["loss"]) < 1e-7
            for n, p in model.named_parameters():
                rel = float((sharded[n] - p.grad).norm() / (p.grad.norm() + 1e-12))
                assert rel < 1e-4, (n, rel)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    except Exception:
        import traceback
        open(os.path.join(out_dir, f"err{rank}"), "w").write(traceback.format_exc())
        raise


def test_sharded_finetune_step_perturb_true_two_ranks_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_perturb_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    errs = [open(tmp_path / f).read() for f in os.listdir(tmp_path) if f.startswith("err")]
    assert not errs, errs[0]
    assert sorted(f for f in os.listdir(tmp_path)) == ["ok0", "ok1"]


def _nccl_worker(rank, world, port, out_dir):
    """One rank per GPU over RCCL (the production backend): the sharded frame equals the single-process frame bit for bit and the
    sharded fine-tune step's all-reduced gradients equal the single-process step's."""
    import numpy as np
    import torch.distributed as dist
    try:
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
        from nerfart_amd import scene, rend_util, dist as nd
        from nerfart_amd.trainer import Trainer
        model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision="bf16x3")
        H, W = 96, 54
        c2w, K = scene.camera(H, W)
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        kw = {k: v for k, v in rk.items() if k != "rayschunk"}
        keys = ("rgb", "depth_volume", "mask_volume", "normals_volume")
        frame = nd.render_sharded(render_fn, o, d, keys=keys, tile=256, detailed_output=False, require_nablas=True, calc_normal=True, **kw)
        om, dm = nd.shard_rays(o, d, tile=256)
        frame2 = nd.render_sharded(render_fn, om, dm, keys=keys, tile=256, n_rays=H * W, detailed_output=False, require_nablas=True,
                                   calc_normal=True, **kw)
        with torch.no_grad():
            _, _, ex = render_fn(o, d, detailed_output=False, require_nablas=True, calc_normal=True, **kw)
        for k in keys:
            np.testing.assert_array_equal(frame[k].cpu().numpy(), ex[k].cpu().numpy(), err_msg=k)
            np.testing.assert_array_equal(frame2[k].cpu().numpy(), ex[k].cpu().numpy(), err_msg=k + " (pre-sharded rays)")
        target = (torch.rand(1, H * W, 3, generator=torch.Generator().manual_seed(2)) * 0.2 + 0.6).to(dev)
        loss_fn = lambda pred, gt: ((pred - gt) ** 2).mean()
        tr = Trainer(model, pass2_rays=64, patches_per_launch=2)
        model.zero_grad()
        out = tr.finetune_step(render_fn, o, d, target, H, loss_fn, **kw)
        sharded = {n: p.grad.clone() for n, p in model.named_parameters()}
        torch.cuda.synchronize()
        dist.barrier()
        # the single-process step on every rank (no process group in the way: world_size() is read from torch.distributed)
        dist.destroy_process_group()
        model.zero_grad()
        ref = Trainer(model, pass2_rays=64, patches_per_launch=2).finetune_step(render_fn, o, d, target, H, loss_fn, **kw)
        assert abs(out["loss"] - ref["loss"]) < 1e-7
        for n, p in model.named_parameters():
            rel = float((sharded[n] - p.grad).norm() / (p.grad.norm() + 1e-12))
            assert rel < 1e-4, (n, rel)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    except Exception:
        import traceback
        open(os.path.join(out_dir, f"err{rank}"), "w").write(traceback.format_exc())
        raise


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: the RCCL path (one process per GPU); the 1-GPU box runs the gloo variant above")
def test_sharded_render_and_finetune_step_over_rccl(tmp_path):
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 8)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # bounded: a collective that never completes must fail this test, not hang the GPU tier (the workers are killed by PID)
    import time
    ctx = mp.spawn(_nccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=False)
    deadline = time.time() + 600
    done = False
    try:
        while not done and time.time() < deadline:
            done = ctx.join(timeout=5)
    finally:
        if not done:
            for pr in ctx.processes:
                if pr.is_alive():
                    pr.kill()
    errs = [open(tmp_path / f).read() for f in os.listdir(tmp_path) if f.startswith("err")]
    assert not errs, errs[0]
    assert done, "the RCCL workers did not finish within 10 minutes"
    assert len([f for f in os.listdir(tmp_path) if f.startswith("ok")]) == world


def test_bench_gpus_2_without_a_launcher_prints_one_parsable_line():
    """`python bench.py --gpus 2 --steps 1` exactly as the driver's N = 1 command is shaped, no torchrun around it: bench.py spawns the two
    ranks itself (tests/test_bench_launcher.py: the decision), rank 0 prints ONE JSON line that says how many ranks the backend saw.
    gloo on one GPU = a functional run of the N-rank code path (RCCL refuses two ranks on one device), not a measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NERFART_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-secondary"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["ranks_seen"] == 2 and js["backend"].startswith("gloo") and len(js["devices"]) == 2
    assert js["steps"] == 1 and js["value"] > 0 and js["scaling"] == "weak"
    assert js["config"]["rays_per_step_per_gpu"] == 480 * 270


def test_bench_refuses_more_rccl_ranks_than_devices():
    """`python bench.py --gpus N` over RCCL with fewer than N visible devices: rc 2 and a message - not an assertion trace, not a hang."""
    import subprocess
    import sys
    n = torch.cuda.device_count() + 1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NERFART_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and f"needs {n} visible devices" in r.stderr and "Traceback" not in r.stderr, (r.returncode, r.stderr[-1000:])
