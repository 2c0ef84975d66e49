#!/usr/bin/env python
"""bench.py - rays/s of the VolSDF render hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" renders one 480 x 270 frame (129,600 rays, 128 coarse + 64 fine samples per ray, up to 6
error-bounded up-sampling rounds) of the synthetic VolSDF scene (nerf-art_amd/scene.py: dims of
configs/volsdf_fangzhou_nature.yaml, seed 0, geometric init + 2% SDF perturbation, beta = 0.01, radiance
gain 4) with rays, weights and workspaces already resident in HBM.  With N ranks every rank renders its
own view per step (views of a camera orbit round-robin over ranks - how the reference's 90-view render
shards) and the rendered tiles are all-gathered over RCCL: weak scaling, value = all rays of all ranks /
max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (fused encode + SDF MLP): achieved =
algorithmic flops per launch (F_sdf = 1,049,088 per point) / average launch duration from HIP
events recorded on the launching stream during the timed steps.  --precision bf16x3 (default): k_sdf_only_bf16,
fp32 operands split into two bf16 terms, three v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate (error
~2^-17, parity tests at 1e-3 like fp32); peak = 2,500 TFLOP/s dense bf16, of which a 3-MFMA product can reach
1/3.  --precision fp32: k_sdf_only on v_mfma_f32_16x16x4_f32 (exact fp32 products), peak = 157.3 TFLOP/s.  `cpu_baseline` times the CPU oracle (a
PyTorch port of the reference algorithm; kind "port") on a strided subset of the same frame's rays.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F_SDF, F_NABLA, F_RAD = 1049088, 918016, 530432          # algorithmic flops / point
H, W, N_SAMPLES, N_IMPORTANCE = 480, 270, 128, 64
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--beta", type=float, default=0.01)
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default="bf16x3",
                    help="fp32: exact v_mfma_f32_16x16x4_f32; bf16x3: split-bf16 operands on v_mfma_f32_16x16x32_bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=768)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # NERFART_BENCH_BACKEND=gloo: functional check of the N > 1 path on a box with fewer GPUs than ranks (every rank on
        # a visible device, collectives staged through the host by nerfart_amd.dist); the measured configuration is RCCL
        backend = os.environ.get("NERFART_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(local)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)

    from nerfart_amd import scene, rend_util, hip, dist as nd

    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision=args.precision)
    kw = {k: v for k, v in rk.items() if k != "rayschunk"}
    n_views = args.warmup + args.steps
    angles = scene.spiral(max(90, n_views * world))
    rays = []
    for s in range(n_views):                      # view of (step s, rank r) = orbit pose s * world + r
        c2w, K = scene.camera(H, W, angle=angles[(s * world + rank) % len(angles)])
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        rays.append((o, d))
    model.packed()
    torch.cuda.synchronize()

    def step(s, detailed=False):
        o, d = rays[s]
        rgb, depth, ex = render_fn(o, d, require_nablas=True, calc_normal=True, detailed_output=detailed, **kw)
        if world > 1:
            tile = torch.cat([rgb[0], depth[0, :, None], ex["normals_volume"][0]], dim=-1)      # [rays, 7]
            nd.all_gather_tiles(tile)
        return rgb, ex

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    hip.profile_begin()
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        step(s)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = hip.profile_end()
    if dist is not None:
        tmax = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)

    rays_per_step = H * W * world
    value = rays_per_step * args.steps / dt

    # algorithmic work of one of this rank's frames (uses the iter_usage the renderer reports)
    _, ex = step(args.warmup, detailed=True)
    torch.cuda.synchronize()
    usage = ex["iter_usage"][0]
    rounds = torch.where(usage < 0, torch.full_like(usage, float(kw["max_upsample_steps"])), usage)
    n_init = 4 * N_SAMPLES
    flops_frame = float((n_init * (1 + rounds) * F_SDF).sum()) + H * W * (N_SAMPLES + N_IMPORTANCE) * (F_SDF + F_NABLA + F_RAD)
    hist = {str(int(k)): int(v) for k, v in zip(*[t.tolist() for t in torch.unique(usage, return_counts=True)])}

    ms, launches, points = prof["k_sdf_only"]
    roofline = None
    if launches > 0:
        flops_per_launch = points / launches * F_SDF
        avg_s = ms / launches * 1e-3
        achieved = flops_per_launch / avg_s / 1e12
        # bf16x3 issues 3 bf16 MFMAs per algorithmic product: priced against the dense bf16 MFMA peak
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
        kname = "k_sdf_only" if args.precision == "fp32" else "k_sdf_only_bf16"
        roofline = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                    "launches": int(launches), "avg_launch_ms": round(ms / launches, 4),
                    "points_per_launch": int(points / launches),
                    "flops_per_point": F_SDF}
        if args.precision == "bf16x3":
            # every algorithmic product is three bf16 MFMAs (hi.hi + hi.lo + lo.hi): the matrix cores do 3x `achieved`
            roofline["mfma_hw_frac"] = round(3.0 * achieved / peak, 4)
        # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs; (2*FETCH + WRITE) KiB with the gfx950 FETCH_SIZE correction) - bench.py itself cannot
        # read hardware counters.
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
        if pm:
            try:
                kk = json.load(open(pm[-1]))["kernels"][kname]
                roofline["traffic"] = int(kk["hbm_bytes_corrected_per_launch"])
                roofline["traffic_source"] = os.path.basename(pm[-1])
                roofline["mfma_util_pmc"] = round(kk.get("mfma_util", 0.0), 4)
            except Exception:
                pass
    kernels_ms = {k: round(v[0] / args.steps, 3) for k, v in prof.items()}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import render as orender
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        n_cpu = args.cpu_rays
        sel = torch.arange(0, H * W, (H * W) // n_cpu)[:n_cpu]
        o, d = rays[args.warmup]
        oc, dc = o[0, sel].cpu(), d[0, sel].cpu()
        cores = torch.get_num_threads()
        with torch.no_grad():
            t1 = time.perf_counter()
            orender.volsdf_render(sd, oc, dc, near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=N_SAMPLES,
                                  N_importance=N_IMPORTANCE, max_upsample_steps=kw["max_upsample_steps"], chunk=256)
            tc = time.perf_counter() - t1
        cpu = {"value": round(n_cpu / tc, 1), "unit": "rays/s", "cores": cores, "kind": "port",
               "sample": f"{n_cpu} rays strided over the same 480x270 frame, {N_SAMPLES}+{N_IMPORTANCE} spp, "
                         f"oracle/render.py volsdf_render on torch-CPU fp32, {tc:.1f} s"}

    if rank == 0:
        out = {
            "metric": "rays/sec at 480x270x128spp VolSDF render",
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16x3 (f32 split into 2 bf16 terms, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "configs[1]: volsdf_fangzhou_nature.yaml dims, 480x270 rays/frame, 128 coarse + 64 fine "
                                   "spp, pure renderer (no CLIP), synthetic random-weight scene beta=%g" % args.beta,
                       "rays_per_step_per_gpu": H * W, "samples_per_ray": N_SAMPLES + N_IMPORTANCE,
                       "parallelism": f"views round-robin over {world} rank(s), all_gather of [rays,7] tiles" if world > 1 else "1 GPU",
                       "samples_per_sec": round(value * (N_SAMPLES + N_IMPORTANCE), 1),
                       "iter_usage_hist": hist,
                       "algorithmic_tflop_per_frame": round(flops_frame / 1e12, 2),
                       "end_to_end_tflops": round(flops_frame * world * args.steps / dt / 1e12, 2),
                       "mlp_kernel_ms_per_step": kernels_ms,
                       "ref_3090_rays_per_s": 6480},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
