#!/usr/bin/env python
"""bench.py - rays/s of the VolSDF render hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1 without a launcher: re-runs itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" renders one 480 x 270 frame (129,600 rays, 128 coarse + 64 fine samples per ray, up to 6
error-bounded up-sampling rounds) of the synthetic VolSDF scene (nerfart_amd/scene.py: dims of
configs/volsdf_fangzhou_nature.yaml, seed 0, geometric init + 2% SDF perturbation, beta = 0.01, radiance
gain 4) with rays, weights and workspaces already resident in HBM.  With N ranks every rank renders its
own view per step (views of a camera orbit round-robin over ranks - how the reference's 90-view render
shards) and the rendered tiles are all-gathered over RCCL: weak scaling, value = all rays of all ranks /
max-over-ranks time.

With N > 1 the same run ALSO times the strong-scaling mode (`"strong"` in the JSON; --shard tiles makes it the
primary line): ONE frame per step cut into 2,048-ray tiles dealt round-robin over the ranks (nerfart_amd.dist.render_sharded,
how a single 960 x 540 frame of cfg 5 is sharded) + one all_gather, value = rays of that one frame / max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (fused encode + SDF MLP, k_sdf_only): achieved =
algorithmic flops per launch (F_sdf = 1,049,088 per point) / average launch duration from HIP
events recorded on the launching stream during the timed steps; peak = 2,500 TFLOP/s (dense bf16 = dense fp16 MFMA).
--precision mixed (default; the mode get_model ships since round 5): every value that reaches a pixel - sdf, nabla, radiance and
compositing of the 192 final samples - in split-bf16 (fp32 operands split into two bf16 terms, three v_mfma_f32_16x16x32_bf16 per product,
fp32 accumulate, error ~2^-17); VolSDF's Algorithm-1 sampler (512 (1 + rounds) no-gradient SDF queries per ray, volsdf.py:479) on the 2-MFMA
kernels (ONE fp16 activation term x fp16 hi + lo weights, v_mfma_f32_16x16x32_f16).  Admitted as the headline by VERDICT r4 "next 2" after it
passed every reference-golden assertion the pure split-bf16 mode passes (tests/test_gpu_bf16x3.py, tests/test_gpu_configs.py, both parametrised
over the sampler; profiles/r06*_mixed_mode_battery.log).  --precision bf16x3: split-bf16 everywhere, the headline of rounds 2-4, now
`secondary.bf16x3` with its own k_sdf_only roofline.  --precision fp32: k_sdf_only on v_mfma_f32_16x16x4_f32 (exact fp32 products), peak =
157.3 TFLOP/s.  `cpu_baseline` times the CPU oracle (a PyTorch port of the reference algorithm; kind "port") on a strided subset of the same
frame's rays, and doubles as a parity check of THIS run (the same rays through the HIP renderer at the benchmarked precision and at fp32-exact,
converged and never-converged rays apart).

`secondary` (N = 1; never the primary `value`): fp32_exact (5 frames), <precision>_vs_fp32_pixels, bf16x3 (pure; 3 frames + pixel statistics +
roofline), fp16x2 (C-ABI precision 4 everywhere, a measurement variant), cfg5_frame_960x540 and cfg4_neus_480x270 (3 frames on 3 views each),
cfg3_finetune_step (BASELINE configs[2]: 1 warm-up + 3 timed steps at perturb=False, stage split, peak memory, the roofline of the dominant
pass-2 kernel k_wgrad<256> from the library's event records; `perturb_true`: the same step with the reference's default render_kwargs_train,
where pass 2 runs the sampler again).
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
# MACs per point the matrix cores actually execute in k_sdf_only[_bf16]: 8 hidden layers with k padded to whole 32-slot units
# (layer 0: 2 encoding units, skip layer 4: 7 + 2 units), the 257-row last layer is a 1-row VALU dot (not MFMA work)
MFMA_MAC_SDF = 256 * (64 + 3 * 256 + 288 + 3 * 256)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0


def cached_n1_line():
    """The newest committed single-GPU bench line under profiles/ (rNN*_bench_line.json with n_gpus == 1 on the 480x270 metric): what an N > 1 line
    quotes as its N = 1 reference (value, cpu_baseline) - the contract measures the CPU baseline on rank 0 at N = 1 only.  (name, dict) or (None, None)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if isinstance(d, dict) and d.get("n_gpus") == 1 and "480x270" in str(d.get("metric", "")) and d.get("value"):
            return os.path.relpath(f, ROOT), d
    return None, None


def scale_fields(world: int, primary_tiles: bool, value: float, ms_per_step: float, secondary: dict, rank_step_ms: list, n1_name, n1_line):
    """The fields that make an N > 1 line readable on its own - pure function of numbers, tested on the CPU:
      strong        the ONE-frame-over-N-GPUs figure (north_star's "ray-parallel scaling"), whichever mode is primary: value, ms_per_step and
                    `speedup_vs_cached_n1` = its rate / the newest committed N = 1 line's (a HINT - the driver computes efficiency itself);
      weak          the views-per-rank figure likewise;
      rank_step_ms  max / mean / min over the ranks of the timed region per step (tile or view imbalance: the max is what `value` is made of);
      cpu_baseline  at N > 1: an explicit marker naming the cached N = 1 figure instead of null."""
    if world == 1:
        return {}
    sec = secondary or {}
    other = sec.get("weak_views" if primary_tiles else "strong_tiles") or {}
    mine = {"value": round(value, 1), "unit": "rays/s", "ms_per_step": round(ms_per_step, 2)}
    strong = dict(mine if primary_tiles else {k: other.get(k) for k in ("value", "unit", "ms_per_step")}, is_primary=bool(primary_tiles),
                  what=f"ONE 480x270 frame per step sharded over {world} ranks in 2,048-ray tiles + one all_gather (strong scaling)")
    weak = dict({k: other.get(k) for k in ("value", "unit", "ms_per_step")} if primary_tiles else mine, is_primary=not primary_tiles,
                what=f"one view per rank per step, {world} frames per step (weak scaling)")
    n1 = float(n1_line["value"]) if n1_line else None
    for d in (strong, weak):
        d["speedup_vs_cached_n1"] = round(d["value"] / n1, 3) if (n1 and d.get("value")) else None
    strong["efficiency_vs_n1_hint"] = round(strong["speedup_vs_cached_n1"] / world, 4) if strong["speedup_vs_cached_n1"] else None
    t = [float(x) for x in rank_step_ms]
    out = {"strong": strong, "weak": weak,
           "rank_step_ms": {"max": round(max(t), 2), "mean": round(sum(t) / len(t), 2), "min": round(min(t), 2), "imbalance_max_over_mean": round(max(t) / (sum(t) / len(t)), 4)},
           "n1_reference": None if not n1_line else {"from": n1_name, "value": n1, "unit": "rays/s", "note": "cached line of an earlier single-GPU run, not measured in this job"},
           "cpu_baseline": {"value": None, "n/a at N>1": True, "unit": "rays/s",
                            "note": "the CPU baseline is timed on rank 0 at N = 1 only (bench contract)",
                            "cached_n1": None if not (n1_line and n1_line.get("cpu_baseline")) else
                            {k: n1_line["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample")}}}
    return out


def launch_plan(n_gpus: int, env: dict, n_devices: int, argv: list, port: int = None):
    """What `python bench.py --gpus N` has to do before anything else.  Returns one of
      ("run", None)        this process is a rank (WORLD_SIZE set by a launcher) or N = 1: go on;
      ("refuse", message)  the RCCL backend needs one visible device per rank and there are fewer (or WORLD_SIZE contradicts --gpus);
      ("spawn", cmd)       N > 1 without a launcher: re-run THIS command line under torch.distributed.run, one process per GPU, rendezvous
                           on 127.0.0.1 (the container hostname may not resolve); rank 0 of that job prints the one JSON line.
    The driver may call `python3 bench.py --gpus 8` directly or through torch.distributed.run: both give the same job."""
    backend = env.get("NERFART_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" in env:
        if int(env["WORLD_SIZE"]) != n_gpus:
            return "refuse", f"--gpus {n_gpus} but the launcher set WORLD_SIZE={env['WORLD_SIZE']}"
        if backend == "nccl" and n_gpus > 1 and n_devices < n_gpus:
            return "refuse", f"--gpus {n_gpus} over RCCL needs {n_gpus} visible devices, this node shows {n_devices}"
        return "run", None
    if n_gpus <= 1:
        return "run", None
    if backend == "nccl" and n_devices < n_gpus:
        return "refuse", (f"--gpus {n_gpus} over RCCL needs {n_gpus} visible devices, this node shows {n_devices} "
                          "(NERFART_BENCH_BACKEND=gloo runs the N-rank path functionally on fewer devices; it is not a measurement)")
    if port is None:
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return "spawn", cmd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--beta", type=float, default=0.01)
    ap.add_argument("--precision", choices=["mixed", "fp32", "bf16x3", "fp16x2"], default="mixed",
                    help="mixed (the headline since round 5 = get_model's default): every value that reaches a pixel in split-bf16 (3 MFMAs per product), "
                         "VolSDF's Algorithm-1 sampler on the 2-MFMA fp16 kernels; bf16x3: split-bf16 everywhere (the headline of rounds 2-4, kept as "
                         "`secondary.bf16x3`); fp32: exact v_mfma_f32_16x16x4_f32; fp16x2: the 2-MFMA form EVERYWHERE (C-ABI precision 4) as the primary "
                         "of an EXPERIMENT line (tools/power_probe_precision.sh) - never the driver's")
    ap.add_argument("--shard", choices=["views", "tiles"], default="views",
                    help="N > 1: views = one view per rank per step (weak scaling, the primary line); tiles = one frame per step "
                         "sharded over the ranks in 2,048-ray tiles (strong scaling).  The other mode is reported as a secondary object.")
    ap.add_argument("--no-calibrate", action="store_true", help="--precision mixed: render with the mode as get_model ships it (2-MFMA sampler), i.e. without the "
                    "render.py line model.calibrate_sampler() (INTEGRATION.md section A); the default runs that line, as render.py does")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (fp32 mode, the other sharding mode)")
    ap.add_argument("--cpu-rays", type=int, default=2048)
    ap.add_argument("--frame", default="480x270", help="HxW of a frame.  480x270 is BASELINE.json's metric (the default and the only size "
                    "the driver's line is quoted on); 960x540 is cfg 5's frame: `--frame 960x540 --steps 90 --shard tiles` is its 90-view loop")
    args = ap.parse_args()
    global H, W
    H, W = (int(v) for v in args.frame.lower().split("x"))

    what, arg = launch_plan(args.gpus, dict(os.environ), torch.cuda.device_count(), sys.argv[1:])
    if what == "refuse":
        print(f"bench.py: {arg}", file=sys.stderr, flush=True)
        sys.exit(2)
    if what == "spawn":
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(arg, env=env))

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
    dev = torch.device("cuda", local)
    # what the collective backend actually sees (goes into the JSON line: a curve point is only as good as its rank count)
    ranks_seen = dist.get_world_size() if dist is not None else 1
    my_dev = f"{torch.cuda.get_device_name(dev)} (cuda:{local})"
    if dist is not None:
        devices = [None] * ranks_seen
        dist.all_gather_object(devices, my_dev)
    else:
        devices = [my_dev]
    if ranks_seen != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the process group has {ranks_seen} ranks", file=sys.stderr, flush=True)
        sys.exit(2)

    from nerfart_amd import scene, rend_util, hip, bench_util, dist as nd

    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision=args.precision)
    # the workload is render.py's (configs[1]): the weights are a checkpoint's and do not change - INTEGRATION.md section A's render.py calls
    # model.calibrate_sampler() after load_state_dict (Algorithm 1's no-gradient SDF queries on the 1-MFMA kernel over error-compensated one-term weights,
    # same guard; nerfart_amd/calibrate.py, ~2 s on the host, once per set of weights: timed here, outside the steps as any checkpoint loading is)
    calibrated, t_cal = False, None
    if args.precision == "mixed" and not args.no_calibrate:
        torch.cuda.synchronize(); t_c0 = time.perf_counter()
        model.calibrate_sampler()
        model.packed_sampler()
        torch.cuda.synchronize(); t_cal = time.perf_counter() - t_c0
        calibrated = model.sampler_precision == "fp16x1c"
    # render_kwargs_test AS get_model BUILDS THEM, `rayschunk` = val_rayschunk (1024) included: every call below is
    # render_fn(rays_o, rays_d, ..., **render_kwargs_test), the reference's own call shape (render.py:527, train.py:189).  volsdf.launch_rays reads the
    # value as the memory hint it is (results are chunk-invariant bit for bit); `secondary.as_the_reference_calls_it` times render.py's 2048 too and
    # the exact honouring (honor_rayschunk=True) beside them
    kw = dict(rk)
    model.render_stats = {}
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

    shared = []                                      # strong mode: every rank holds the SAME frame's rays (rank 0's views)
    for s_ in range(n_views):
        c2w, K = scene.camera(H, W, angle=angles[(s_ * world) % len(angles)])
        o_, d_, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        shared.append((o_, d_))

    # ... of which each rank keeps its own 2,048-ray tiles resident (the inputs of the timed step: the shard's rays, like the whole
    # frame's rays in the other mode, are made before the clock starts - nerfart_amd.dist.shard_rays)
    shared_mine = [nd.shard_rays(o_, d_, tile=2048) for (o_, d_) in shared] if world > 1 else shared

    def step_tiles(s, detailed=False):
        o, d = shared_mine[s]
        return nd.render_sharded(render_fn, o, d, tile=2048, n_rays=H * W, require_nablas=True, calc_normal=True, detailed_output=False, **kw)

    rank_times = {}

    def timed(fn, profile=False):
        """W warm-up calls, then exactly K timed ones between barrier + synchronize pairs; max over ranks."""
        for s_ in range(args.warmup):
            fn(s_)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if profile:
            hip.profile_begin()
        t0 = time.perf_counter()
        for s_ in range(args.warmup, args.warmup + args.steps):
            fn(s_)
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0              # this rank's own K steps (its collectives included), before it waits for the others
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        prof_ = hip.profile_end() if profile else None
        if dist is not None:
            # max over ranks = the contract's time; the per-rank times before the closing barrier show tile / view imbalance
            tall = [None] * dist.get_world_size()
            dist.all_gather_object(tall, (dt_, t_own))
            rank_times[fn.__name__] = [t_[1] / args.steps * 1e3 for t_ in tall]
            dt_ = max(t_[0] for t_ in tall)
        return dt_, prof_

    primary_tiles = args.shard == "tiles" and world > 1
    dt, prof = timed(step_tiles if primary_tiles else step, profile=True)

    rays_per_step = H * W * (1 if primary_tiles else world)
    value = rays_per_step * args.steps / dt

    # secondary measurements (never the primary `value`): the other sharding mode at N > 1; the exact-fp32 mode at N = 1
    secondary = {}
    if world > 1 and not args.no_secondary:
        dt2, _ = timed(step if primary_tiles else step_tiles)
        r2 = H * W * (world if primary_tiles else 1)
        secondary["weak_views" if primary_tiles else "strong_tiles"] = {
            "value": round(r2 * args.steps / dt2, 1), "unit": "rays/s", "ms_per_step": round(dt2 / args.steps * 1e3, 2),
            "scaling": "weak" if primary_tiles else "strong",
            "what": ("one view per rank per step" if primary_tiles else
                     "ONE 480x270 frame per step, 2,048-ray tiles round-robin over the ranks (dist.render_sharded) + one all_gather")}
        # cfg 5's frame size: ONE 960 x 540 frame (518,400 rays) sharded the same way, one warm-up and one timed frame
        H5, W5 = 960, 540
        c2w5, K5 = scene.camera(H5, W5, angle=angles[0])
        o5, d5, _ = rend_util.get_rays(c2w5[None].to(dev), K5[None].to(dev), H5, W5)
        o5m, d5m = nd.shard_rays(o5, d5, tile=2048)
        big = lambda: nd.render_sharded(render_fn, o5m, d5m, tile=2048, n_rays=H5 * W5, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        big()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t5 = time.perf_counter()
        big()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t5 = torch.tensor([time.perf_counter() - t5], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t5, op=dist.ReduceOp.MAX)
        secondary["strong_tiles_960x540"] = {"value": round(H5 * W5 / float(t5), 1), "unit": "rays/s", "ms_per_step": round(float(t5) * 1e3, 2), "steps": 1,
                                             "scaling": "strong", "what": "ONE 960x540 frame (cfg 5), 2,048-ray tiles round-robin over the ranks + one all_gather"}
    headline_split = args.precision in ("mixed", "bf16x3")
    other = "bf16x3" if args.precision == "mixed" else "mixed"          # the split-bf16 mode that is NOT the headline: a secondary of its own
    if world == 1 and not args.no_secondary and headline_split:
        m32, _, f32 = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision="fp32")
        m32.packed()
        o_, d_ = rays[0]
        a32, _, _ = f32(o_, d_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        # pixel agreement of the benchmarked precision with the exact-fp32 frame (same view): rays past the north-star 1e-3 are rays
        # whose error-bounded sampling took another branch in the two arithmetics (tests/test_gpu_bf16x3.py bounds their share)
        a16, _, _ = render_fn(o_, d_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        e16 = (a16 - a32).abs().max(dim=-1).values
        pix = {"rays": int(e16.numel()), "rays_over_1e-3": int((e16 > 1e-3).sum()), "max_abs": round(float(e16.max()), 6),
               "p999_abs": round(float(e16.flatten().kthvalue(int(0.999 * e16.numel())).values), 7),
               "psnr_db": round(float(-10 * torch.log10(((a16 - a32) ** 2).mean().clamp_min(1e-20))), 1)}
        # the 2-MFMA variant (C-ABI precision 4: one fp16 activation term x fp16 hi + lo weights) - a SECONDARY, never the headline:
        # the same pixel statistics against the same exact-fp32 frame, and its frame time (filled in below)
        m16, _, f16 = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision="fp16x2")
        b16_, _, _ = f16(o_, d_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        e2 = (b16_ - a32).abs().max(dim=-1).values
        pix2 = {"rays": int(e2.numel()), "rays_over_1e-3": int((e2 > 1e-3).sum()), "max_abs": round(float(e2.max()), 6),
                "p999_abs": round(float(e2.flatten().kthvalue(int(0.999 * e2.numel())).values), 7),
                "psnr_db": round(float(-10 * torch.log10(((b16_ - a32) ** 2).mean().clamp_min(1e-20))), 1)}
        # ... and the other split-bf16 mode (headline mixed: pure bf16x3, the headline of rounds 2-4, so the series stays comparable; headline
        # bf16x3: the mixed mode = Algorithm 1's 512 (1 + rounds) no-gradient SDF queries per ray on the 2-MFMA kernels, the 192 final
        # samples - every number that reaches a pixel - in split-bf16, C entry point nerfart_volsdf_render_mixed_fwd)
        mmx, _, fmx = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision=other)
        bmx, _, _ = fmx(o_, d_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        e3 = (bmx - a32).abs().max(dim=-1).values
        pix3 = {"rays": int(e3.numel()), "rays_over_1e-3": int((e3 > 1e-3).sum()), "max_abs": round(float(e3.max()), 6),
                "p999_abs": round(float(e3.flatten().kthvalue(int(0.999 * e3.numel())).values), 7),
                "psnr_db": round(float(-10 * torch.log10(((bmx - a32) ** 2).mean().clamp_min(1e-20))), 1)}
        del a32, a16, e16, b16_, e2, bmx, e3
        n32 = 5                                           # five timed frames (five views of the orbit), one warm-up above
        views32 = []
        for s_ in range(n32):
            c2w_, K_ = scene.camera(H, W, angle=angles[(7 * s_ + 3) % len(angles)])
            views32.append(rend_util.get_rays(c2w_[None].to(dev), K_[None].to(dev), H, W)[:2])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for oo_, dd_ in views32:
            f32(oo_, dd_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        t32 = (time.perf_counter() - t1) / n32
        secondary["fp32_exact"] = {"value": round(H * W / t32, 1), "unit": "rays/s", "ms_per_step": round(t32 * 1e3, 2), "steps": n32,
                                   "what": "same workload with --precision fp32 (v_mfma_f32_16x16x4_f32, exact fp32 products; reverse-mode grad(SDF) kernel)",
                                   "vs_ref_3090": round(H * W / t32 / 6480.0, 2)}
        secondary[f"{args.precision}_vs_fp32_pixels"] = pix
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for oo_, dd_ in views32[:3]:
            f16(oo_, dd_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        t16 = (time.perf_counter() - t1) / 3
        secondary["fp16x2"] = {"value": round(H * W / t16, 1), "unit": "rays/s", "ms_per_step": round(t16 * 1e3, 2), "steps": 3,
                               "vs_fp32_pixels": pix2,
                               "what": "same workload at C-ABI precision 4: ONE fp16 activation term x fp16 hi + lo weights, 2 x v_mfma_f32_16x16x32_f16 per "
                                       "product (11-bit activations, TF32 class) - a measurement variant, NOT the headline precision; table vs the "
                                       "oracle: profiles/r12_parity_table.json (tools/parity_table.py)"}
        torch.cuda.synchronize()
        hip.profile_begin()
        t1 = time.perf_counter()
        for oo_, dd_ in views32[:3]:
            fmx(oo_, dd_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        tmx = (time.perf_counter() - t1) / 3
        pmx = hip.profile_end()
        k_ms, k_n, k_pts = pmx["k_sdf_only"]
        secondary["bf16x3" if other == "bf16x3" else "bf16x3_with_fp16x2_sampler"] = {
            "value": round(H * W / tmx, 1), "unit": "rays/s", "ms_per_step": round(tmx * 1e3, 2), "steps": 3, "vs_fp32_pixels": pix3,
            # the dominant kernel of THAT mode, priced like `roofline` (algorithmic flops / launch time / 2,500 TFLOP/s)
            "roofline_k_sdf_only": None if not k_n else {
                "kernel": "k_sdf_only_bf16" if other == "bf16x3" else "f16x2::k_sdf_only_bf16", "mfma_per_product": 3 if other == "bf16x3" else 2,
                "achieved": round(k_pts / k_n * F_SDF / (k_ms / k_n * 1e-3) / 1e12, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(k_pts / k_n * F_SDF / (k_ms / k_n * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4), "avg_launch_ms": round(k_ms / k_n, 4),
                "launches": int(k_n)},
            "what": ("split-bf16 EVERYWHERE (model.set_precision('bf16x3')): the headline precision of rounds 2-4 (BENCH_r02..r04), kept so that the "
                     "series stays comparable" if other == "bf16x3" else
                     "Algorithm 1's SDF queries at C-ABI precision 4, the 192 final samples (sdf, nabla, radiance, compositing) in "
                     "split-bf16: model.set_precision('mixed')")}
        if calibrated:
            mgm, _, fgm = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision="mixed")
            fgm(o_, d_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for oo_, dd_ in views32[:3]:
                fgm(oo_, dd_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
            torch.cuda.synchronize()
            tgm = (time.perf_counter() - t1) / 3
            secondary["mixed_as_get_model_ships"] = {
                "value": round(H * W / tgm, 1), "unit": "rays/s", "ms_per_step": round(tgm * 1e3, 2), "steps": 3,
                "what": "the mode exactly as frameworks.get_model returns it - 2-MFMA sampler (fp16 act x fp16 hi + lo weights), same guard - i.e. render.py WITHOUT "
                        "the calibrate_sampler() line, and what pass 1 of a training step runs (its weights change every step)"}
            del mgm, fgm
        # the 1-MFMA sampler on NEAREST-rounded weights (C-ABI precision 5 without the calibration, opt-in, NOT shipped: one or two rays of 2,048 more than pure split-bf16 past 1e-3 on 2 of 8 views,
        # profiles/r09_guard_sweep_fp16x1_8views.json) at the guard where its statistics come closest to the shipped mode's - what 1 MFMA per product buys
        mx1, _, fx1 = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision="bf16x3")
        mx1.set_sampler_precision("fp16x1", guard=0.05)
        mx1.render_stats = {}
        bx1, _, _ = fx1(o_, d_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        hip.profile_begin()
        t1 = time.perf_counter()
        for oo_, dd_ in views32[:3]:
            fx1(oo_, dd_, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        tx1 = (time.perf_counter() - t1) / 3
        px1 = hip.profile_end()
        k_ms, k_n, k_pts = px1["k_sdf_only"]
        secondary["bf16x3_with_fp16x1_sampler_guard_0.05"] = {
            "value": round(H * W / tx1, 1), "unit": "rays/s", "ms_per_step": round(tx1 * 1e3, 2), "steps": 3,
            "rays_sampled_twice_frac": round(mx1.render_stats["escalated"] / max(mx1.render_stats["rays"], 1), 5),
            "roofline_k_sdf_only": None if not k_n else {
                "kernel": "f16x1::k_sdf_only_bf16", "mfma_per_product": 1,
                "achieved": round(k_pts / k_n * F_SDF / (k_ms / k_n * 1e-3) / 1e12, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(k_pts / k_n * F_SDF / (k_ms / k_n * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4), "avg_launch_ms": round(k_ms / k_n, 4),
                "launches": int(k_n)},
            "what": "OPT-IN, not the headline: Algorithm 1's SDF queries on the 1-MFMA kernel (model.set_sampler_precision('fp16x1', guard=0.05); one fp16 "
                    "activation term x one fp16 weight term), the 192 final samples in split-bf16; measured against the shipped mode's contract and short "
                    "of it by one or two rays of 2,048 on 2 of 8 views (DESIGN.md 4.1e)"}
        del m32, f32, m16, f16, mmx, fmx, mx1, fx1, bx1
        # the other single-GPU configurations of BASELINE.json: one warm-up frame, then N_SEC timed frames on N_SEC views of the orbit
        # (bench lines of their own: tools/bench_neus.py, tools/bench_train.py)
        N_SEC = 3

        def frames(fn, Hh, Ww, **extra):
            views = []
            for s_ in range(N_SEC + 1):
                c2w_, K_ = scene.camera(Hh, Ww, angle=angles[(11 * s_ + 1) % len(angles)])
                views.append(rend_util.get_rays(c2w_[None].to(dev), K_[None].to(dev), Hh, Ww)[:2])
            fn(*views[0], calc_normal=True, detailed_output=False, **extra)
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for oo, dd in views[1:]:
                fn(oo, dd, calc_normal=True, detailed_output=False, **extra)
            torch.cuda.synchronize()
            return (time.perf_counter() - t_) / N_SEC
        # the frame through the reference's call shapes, `rayschunk` left in - val_rayschunk 1024 (volsdf.py:990 -> train.py:189,
        # render.py:527; = the primary line's kwargs) and render.py's own default 2048 (render.py:488,614) - and, for the record, what slicing exactly as
        # asked costs (127 / 64 launches per frame, each with its own <= 7 host-synchronised sampler rounds)
        as_ref = {}
        for rc_ in (1024, 2048):
            t_rc = frames(render_fn, H, W, require_nablas=True, **dict(kw, rayschunk=rc_))
            as_ref[f"rayschunk_{rc_}"] = {"value": round(H * W / t_rc, 1), "unit": "rays/s", "ms_per_step": round(t_rc * 1e3, 2), "steps": N_SEC,
                                          "frac_of_primary": round(H * W / t_rc / value, 4)}
        oo_, dd_ = rays[0]
        render_fn(oo_, dd_, require_nablas=True, calc_normal=True, detailed_output=False, **dict(kw, rayschunk=2048, honor_rayschunk=True))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        render_fn(oo_, dd_, require_nablas=True, calc_normal=True, detailed_output=False, **dict(kw, rayschunk=2048, honor_rayschunk=True))
        torch.cuda.synchronize()
        t_h = time.perf_counter() - t1
        as_ref["rayschunk_2048_honoured_exactly"] = {"value": round(H * W / t_h, 1), "unit": "rays/s", "ms_per_step": round(t_h * 1e3, 2), "steps": 1}
        as_ref["what"] = ("render_fn(rays_o, rays_d, ..., **render_kwargs_test) with the `rayschunk` key left in, 3 frames each: the library reads it as the "
                          "reference's 24 GB memory hint and launches max(rayschunk, 131,072) rays (volsdf.launch_rays; bit-identical results, "
                          "tests/test_gpu_configs.py); honor_rayschunk=True slices exactly as asked")
        secondary["as_the_reference_calls_it"] = as_ref
        t5 = frames(render_fn, 960, 540, require_nablas=True, **kw)
        secondary["cfg5_frame_960x540"] = {"value": round(960 * 540 / t5, 1), "unit": "rays/s", "ms_per_step": round(t5 * 1e3, 2), "steps": N_SEC,
                                           "what": "configs[4] frame size (518,400 rays, VolSDF 128 + 64 spp) on ONE GPU"}
        mn, rkn, fn_n = scene.build_model("NeuS", seed=0, beta=None, device=dev, precision=args.precision)
        t4 = frames(fn_n, H, W, **rkn)                  # rayschunk = val_rayschunk (512) left in, as neus.py:747 builds the kwargs
        F_NEUS = 128 * F_SDF + 128 * (F_SDF + F_NABLA) + 127 * (F_SDF + F_NABLA + 542720)          # 704.8 MFLOP per ray
        secondary["cfg4_neus_480x270"] = {"value": round(H * W / t4, 1), "unit": "rays/s", "ms_per_step": round(t4 * 1e3, 2), "steps": N_SEC,
                                          "end_to_end_tflops": round(H * W * F_NEUS / t4 / 1e12, 1),
                                          "frac_of_bf16_peak_end_to_end": round(H * W * F_NEUS / t4 / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                                          "what": "configs[3]: neus_fangzhou_vangogh.yaml dims, 64 + 64 spp, 704.8 MFLOP/ray algorithmic; "
                                                  "kernel-level evidence: profiles/r05*_neus_kernel_stats.txt"}
        del mn, fn_n
        # BASELINE configs[2] (SURVEY cfg 3): the fine-tune step - HIP pass 1 with kept state, CLIP + VGG style losses on the hand-written
        # kernels, native pass 2 (nerfart_volsdf_render_bwd per launch group), Adam.  One warm-up + N_SEC timed steps; the dominant
        # pass-2 kernel (k_wgrad<256>, HBM bound) priced from the library's own event records of these steps.
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        ctx3 = bench_util.finetune_setup(dev, H, W, beta=args.beta, angle=angles[2], precision=args.precision)
        m3, loss3, eik3, prof3 = bench_util.finetune_steps(ctx3, N_SEC, warmup=1, profile=True)
        wg_ms, wg_n, wg_bytes = prof3["k_wgrad256"]
        secondary["cfg3_finetune_step"] = {
            "value": round(sum(m3), 4), "unit": "s/step", "higher_is_better": False, "steps": N_SEC, "rays_per_s": round(H * W / sum(m3), 1),
            "pass1_render_s": round(m3[0], 4), "style_losses_fwd_bwd_s": round(m3[1], 4), "pass2_render_bwd_s": round(m3[2], 4), "adam_s": round(m3[3], 4),
            "loss": round(loss3, 5), "eikonal": round(float(eik3), 7), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            "roofline_pass2_dominant": None if wg_n == 0 else {
                "bound": "hbm", "kernel": "k_wgrad<256>", "achieved": round(wg_bytes / (wg_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(wg_bytes / (wg_ms * 1e-3) / 8e12, 4), "launches_per_step": int(wg_n // N_SEC), "ms_per_step": round(wg_ms / N_SEC, 2),
                "algorithmic_GB_per_step": round(wg_bytes / N_SEC / 1e9, 1),
                "what": "weight-gradient reductions over the point-major bf16 dumps: both operands read once (DESIGN.md 4.3)"},
            "mlp_kernel_ms_per_step": {k: round(v[0] / N_SEC, 2) for k, v in prof3.items()},
            "what": "configs[2]: volsdf_fangzhou_vangogh.yaml train step at 480x270 (render + CLIP directional / contrastive / PatchNCE + VGG "
                    "perceptual, backward, Adam), seeded random-weight CLIP ViT-B/32 + VGG16, perturb=False (pass 2 reads pass 1's kept samples: at "
                    "perturb=False re-sampling reproduces them), precision %s; `perturb_true` = the same step with render_kwargs_train as the reference "
                    "builds them" % args.precision}
        # what a weight update costs before the next render: both blobs (+ the mixed mode's sampler blob) re-packed on the C ABI
        # (nerfart_pack_surface_blob / _radiance_blob: fold, permutation, hi / lo split on the device) - part of pass1_render_s above
        with torch.no_grad():
            for p_ in ctx3["model"].parameters():
                p_.add_(0.0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ctx3["model"].packed(); ctx3["model"].packed_sampler()
        torch.cuda.synchronize()
        secondary["cfg3_finetune_step"]["repack_s"] = round(time.perf_counter() - t1, 5)
        # THE REFERENCE'S DEFAULT render_kwargs_train: perturb=True (volsdf.py:982; no shipped YAML overrides it).  Both passes call the renderer
        # (volsdf.py:724-728, :759-766): pass 1 on the fused renderer with random final samples, nothing kept; pass 2 runs Algorithm 1 AGAIN with
        # fresh draws and nerfart_volsdf_render_bwd re-evaluates the per-point state (have_state = 0).  tests: FP_* goldens of the reference Trainer.
        m3p, loss3p, eik3p, _ = bench_util.finetune_steps(ctx3, 2, warmup=1, perturb=True)
        ctx3["trainer"].share_algorithm1 = False          # ... and with a SECOND full run of Algorithm 1 in pass 2 (how round 5 first built it)
        m3q, _, _, _ = bench_util.finetune_steps(ctx3, 2, warmup=1, perturb=True)
        ctx3["trainer"].share_algorithm1 = True
        secondary["cfg3_finetune_step"]["perturb_true"] = {
            "value": round(sum(m3p), 4), "unit": "s/step", "higher_is_better": False, "steps": 2, "rays_per_s": round(H * W / sum(m3p), 1),
            "pass1_render_s": round(m3p[0], 4), "style_losses_fwd_bwd_s": round(m3p[1], 4), "pass2_render_bwd_s": round(m3p[2], 4), "adam_s": round(m3p[3], 4),
            "loss": round(loss3p, 5), "eikonal": round(float(eik3p), 7),
            "with_a_second_algorithm1_run": {"value": round(sum(m3q), 4), "unit": "s/step", "steps": 2, "pass1_render_s": round(m3q[0], 4),
                                             "pass2_render_bwd_s": round(m3q[2], 4), "sampler_alone_s": round(bench_util.pass2_sampler_seconds(ctx3), 4)},
            "what": "render_kwargs_train['perturb'] = True, the reference's default: pass 2 back-propagates through its OWN random final samples.  "
                    "Algorithm 1 runs ONCE (its rounds draw nothing and would repeat exactly: the weights do not change between the passes) with two "
                    "draws per ray - pass 1's fine samples and pass 2's (Trainer.render_two_draws); pass 2 = forward re-evaluation at its samples + "
                    "backward.  with_a_second_algorithm1_run: Trainer(share_algorithm1=False) - the same samples for the same draws, bit for bit (tests).  "
                    "Trainer(reuse_pass1_samples=True) is the opt-in that uses pass 1's samples (INTEGRATION.md section F)"}
        # the same (perturb=False) step in the OTHER split-bf16 mode (no gradient flows through the sampler, volsdf.py:479; the kept state and
        # pass 2 are split-bf16 in both)
        ctx3["model"].set_precision(other)
        m3b, loss3b, _, _ = bench_util.finetune_steps(ctx3, 2, warmup=1)
        secondary["cfg3_finetune_step"]["with_bf16x3_sampler" if other == "bf16x3" else "with_fp16x2_sampler"] = {
            "value": round(sum(m3b), 4), "unit": "s/step", "steps": 2, "pass1_render_s": round(m3b[0], 4), "pass2_render_bwd_s": round(m3b[2], 4), "loss": round(loss3b, 5)}
        del ctx3
        torch.cuda.empty_cache()

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
        # (mixed: every k_sdf_only launch of a frame is Algorithm 1's - the 2-MFMA kernel, namespace f16x2; the final samples run k_sdf_grad_bf16)
        kname = {"fp32": "k_sdf_only", "bf16x3": "k_sdf_only_bf16"}.get(args.precision, "f16x1::k_sdf_only_bf16" if calibrated else "f16x2::k_sdf_only_bf16")
        roofline = {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                    "launches": int(launches), "avg_launch_ms": round(ms / launches, 4),
                    "points_per_launch": int(points / launches),
                    "flops_per_point": F_SDF}
        if args.precision in ("mixed", "fp16x2"):
            # ONE fp16 activation term x fp16 hi + lo weight terms: 2 x v_mfma_f32_16x16x32_f16 per product (the dense fp16 peak = the bf16 one)
            mpp = 1 if calibrated else 2              # (the four k-steps of ready-made encoding units of K2's 59 run three terms in either form)
            roofline["mfma_per_product"] = mpp
            roofline["mfma_executed_frac"] = round(mpp * 2.0 * MFMA_MAC_SDF * (points / launches) / avg_s / 1e12 / peak, 4)
            roofline["ceiling_frac"] = {"matrix_pipe_only": round(0.91 / mpp, 4), "source": "profiles/r02s_ubench_coissue.txt (MFMA-only rate 0.91 of peak)",
                                        "units": "ALGORITHMIC flops, i.e. comparable with `frac` (a stream of nothing but mfma_per_product-MFMA products reaches 0.91 / mfma_per_product); "
                                                 "`mfma_executed_frac` (what the matrix pipe executes: mfma_per_product MFMAs per product + tile padding) compares with 0.91 "
                                                 "(MFMAs alone) and with 0.56 - 0.58 (a synthetic stream of the 3-MFMA kernel's own instruction mix at the power "
                                                 "cap, profiles/r03u_ubench_power.txt)"}
            roofline["note"] = ("the sampler's kernel: Algorithm 1's 512 (1 + rounds) SDF queries per ray, 69 % of a bf16x3 frame; frac = ALGORITHMIC flops "
                                "(F_sdf per point) / launch time / 2,500 TFLOP/s, comparable across modes; the sustained rate is set by the package "
                                "power cap (joules per product: profiles/r05n power probes), which is why fewer MFMAs per product buy time" +
                                ("; calibrated sampler: ONE v_mfma_f32_16x16x32_f16 per product over error-compensated one-term weights (csrc/mlp_chain_f16x1.hip, "
                                 "nerfart_amd/calibrate.py) - with one MFMA per item the kernel's time follows its fragment reads, LDS-DMA pieces and epilogue VALU, not the "
                                 "matrix pipe; it still draws the package limit, at a 7 % higher clock (profiles/r12_power_k2.json; DESIGN.md 4.1e / 4.1f / 5)" if calibrated else ""))
        if args.precision == "bf16x3":
            # what the matrix pipe executes: MFMA_MAC_SDF multiply-adds per point, each as `mfma_per_product` bf16 MFMAs
            # (hi.hi + hi.lo + lo.hi), as a fraction of the dense bf16 peak
            roofline["note"] = ("measured limiters (DESIGN.md 4.1b): in isolation, weight bytes moved per column - 1.9 MB of split-bf16 fragments per "
                                "128-point tile through L2 -> LDS-DMA -> ds_read_b128 (profiles/r02k_ablate_w32.log); in the sustained frame, the "
                                "package power cap: 1.28 - 1.33 kW of 1.4 kW while rendering, shader clock ~2.1 GHz instead of the 2.4 GHz the "
                                "peak assumes (profiles/r03l_power_probe.log) - 3 MFMAs per product at ~0.6 of the dense-bf16 rate")
            # ceilings for this kernel in the same algorithmic-flop units (profiles/r02s_ubench_coissue.txt, tools/ubench_coissue.hip):
            # 3 MFMAs per product at the measured MFMA-only rate (17.7 nominal cycles per 16x16x32 = 0.91 of peak), and the SIMD's
            # measured issue capacity for the kernel's own mix of MFMAs, fragment reads, LDS-DMA pieces and epilogue VALU (25.0 cycles)
            roofline["ceiling_frac"] = {"matrix_pipe_only": round(0.91 / 3, 4), "issue_capacity_for_this_mix": round(0.91 / 3 * 17.7 / 25.0, 4),
                                        "source": "profiles/r02s_ubench_coissue.txt",
                                        # a synthetic stream with this kernel's mix (3 MFMAs + 2 ds_read_b128 + ~6 VALU per item, random
                                        # operands) SUSTAINS 0.562 - 0.577 of the dense-bf16 peak in executed MFMA flops at 1.23 - 1.27 kW;
                                        # MFMAs alone 0.957 at 1.32 kW (tools/ubench_power.hip) - compare with mfma_executed_frac below
                                        "sustained_executed_frac_same_mix": 0.577, "sustained_executed_frac_mfma_only": 0.957,
                                        "source_sustained": "profiles/r03u_ubench_power.txt"}
            roofline["mfma_per_product"] = 3
            roofline["mfma_executed_frac"] = round(3 * 2.0 * MFMA_MAC_SDF * (points / launches) / avg_s / 1e12 / peak, 4)
        # HBM traffic per launch is a PROFILED figure, not measured in this run (bench.py cannot read hardware counters): it
        # comes from the newest committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; (2*FETCH +
        # WRITE) KiB with the gfx950 FETCH_SIZE correction) and is reported as `traffic` only if that profile was taken on
        # the kernel sources this run executes (source hash recorded by tools/pmc_summary.py); otherwise it is marked stale.
        import glob
        # (only profiles of THIS workload: the train-step / CLIP profiles hold the same kernel names at other launch sizes)
        pm = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))
                    if not any(t in os.path.basename(f) for t in ("_train_", "_clip_", "_neus_")))
        if pm:
            try:
                js = json.load(open(pm[-1]))
                kk = js["kernels"][kname]
                if any("bench.py" not in c or "bench_" in c for c in js.get("commands", [])):
                    raise ValueError("not a profile of bench.py")
                from nerfart_amd import volsdf as _volsdf       # (per-launch bytes of the sampler's kernel scale with the rays per chunk)
                fresh = js.get("csrc_sha256") == hip.csrc_sha256() and js.get("default_rayschunk") == _volsdf.DEFAULT_RAYSCHUNK
                roofline["traffic_profiled"] = {"bytes_per_launch": int(kk["hbm_bytes_corrected_per_launch"]), "source": os.path.basename(pm[-1]),
                                                "profile_csrc_sha256": js.get("csrc_sha256"), "matches_this_build": fresh,
                                                "mfma_busy_pmc": round(kk.get("mfma_util", 0.0), 4)}
                roofline["traffic"] = int(kk["hbm_bytes_corrected_per_launch"]) if fresh else None
            except Exception:
                pass
    kernels_ms = {k: round(v[0] / args.steps, 3) for k, v in prof.items() if k != "k_wgrad256"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import render as orender
        try:
            import psutil
            cores = psutil.cpu_count(logical=False) or os.cpu_count()
        except Exception:
            cores = os.cpu_count()
        cpu_model = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "unknown")
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o, d = rays[args.warmup]

        last_ref = {}

        def cpu_run(n):                               # n rays strided over the frame, ONE chunk (>= 2,048-ray chunks once calibrated)
            sel = torch.arange(0, H * W, (H * W) // n)[:n]
            oc, dc = o[0, sel].cpu(), d[0, sel].cpu()
            with torch.no_grad():
                t1 = time.perf_counter()
                ref = orender.volsdf_render(sd, oc, dc, near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=N_SAMPLES,
                                            N_importance=N_IMPORTANCE, max_upsample_steps=kw["max_upsample_steps"], chunk=max(n, 2048))
                dt1 = time.perf_counter() - t1
            last_ref.update(sel=sel, ref=ref)
            return dt1
        # calibration: the port's many small tensor ops do not scale to every core of a 2-socket host - pick the thread count
        # (never more than one per PHYSICAL core) that is fastest on 256 rays, then size the sample to about 20 s of CPU work
        cands = sorted({int(cores), min(int(cores), 32), min(int(cores), 8)}, reverse=True)
        t_cal, threads = None, int(cores)
        for th in cands:
            torch.set_num_threads(th)
            cpu_run(64)                               # warm the thread pool / allocator at this size
            t = cpu_run(256)
            if t_cal is None or t < t_cal:
                t_cal, threads = t, th
        torch.set_num_threads(threads)
        cores = threads
        n_cpu = int(min(max(20.0 * 256 / t_cal, 512), args.cpu_rays * 2) // 256 * 256)      # about 20 s of CPU work
        tc = cpu_run(n_cpu)
        # the timed sample doubles as a parity check of THIS run: the same rays through the HIP renderer against the oracle's result
        sel, ref = last_ref["sel"].to(dev), last_ref["ref"]
        with torch.no_grad():
            g_rgb, g_depth, g_ex = render_fn(o[:, sel], d[:, sel], require_nablas=True, calc_normal=True, detailed_output=True, **kw)
        same = (g_ex["iter_usage"][0].cpu() == ref["iter_usage"])
        e_pix = (g_rgb[0].cpu() - ref["rgb"]).abs().max(dim=-1).values
        conv = same & (ref["iter_usage"] >= 0)            # rays whose error-bounded sampling converged, in the same rounds on both sides
        parity = {"rays": int(sel.numel()), "same_upsampling_rounds_frac": round(float(same.float().mean()), 5),
                  "never_converged_rays_oracle": int((ref["iter_usage"] < 0).sum()),
                  "max_abs_rgb_converged_same_rounds": float(f"{float(e_pix[conv].max()):.3e}"),
                  "rays_over_1e-3_among_converged": int((e_pix[conv] > 1e-3).sum()),
                  "max_abs_rgb_same_rounds": float(f"{float(e_pix[same].max()):.3e}"),
                  "max_abs_rgb_all": float(f"{float(e_pix.max()):.3e}"), "rays_over_1e-3": int((e_pix > 1e-3).sum()),
                  "psnr_db": round(float(-10 * torch.log10(((g_rgb[0].cpu() - ref["rgb"]) ** 2).mean().clamp_min(1e-20))), 1),
                  "max_abs_depth_same_rounds": float(f"{float((g_depth[0].cpu() - ref['depth_volume'])[same].abs().max()):.3e}")}
        # ... held to the SAME statement tests/test_gpu_configs.py::test_cfg2_mixed_mode_over_eight_orbit_views asserts on 8 views (bench_util.view_budget):
        # the view sampled here is orbit pose `warmup` - 5 on the driver's command, 1 at the defaults, both among the tested views
        st_ = bench_util.pixel_stats(g_rgb[0].cpu(), g_ex["iter_usage"][0].cpu(), ref["rgb"], ref["iter_usage"])
        viol_ = bench_util.view_budget(st_)
        parity["rays_over_1e-3_among_oracle_converged"] = st_["rays_over_1e-3_among_oracle_converged"]
        parity["view"] = f"orbit pose {args.warmup % len(angles)} of scene.spiral({len(angles)})"
        parity["within_the_tests_budget"] = not viol_
        if viol_:
            parity["budget_violations"] = viol_
        # the exact-fp32 mode against the oracle on the SAME rays: says whether a ray past 1e-3 is the split-bf16 arithmetic or
        # Algorithm 1's own discontinuities (a ray that flips under any change of rounding; tools/fp32_outlier.py)
        if headline_split and not args.no_secondary:
            m32, _, f32 = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision="fp32")
            with torch.no_grad():
                r32, d32, x32 = f32(o[:, sel], d[:, sel], require_nablas=True, calc_normal=True, detailed_output=True, **kw)
            same32 = (x32["iter_usage"][0].cpu() == ref["iter_usage"])
            e32 = (r32[0].cpu() - ref["rgb"]).abs().max(dim=-1).values
            over16 = (e_pix > 1e-3).nonzero().flatten().tolist()
            secondary.setdefault("fp32_exact", {})["parity_vs_oracle_same_rays"] = {
                "rays": int(sel.numel()), "same_upsampling_rounds_frac": round(float(same32.float().mean()), 5),
                "rays_over_1e-3": int((e32 > 1e-3).sum()), "max_abs_rgb_all": float(f"{float(e32.max()):.3e}"),
                "max_abs_rgb_same_rounds": float(f"{float(e32[same32].max()):.3e}"),
                "psnr_db": round(float(-10 * torch.log10(((r32[0].cpu() - ref["rgb"]) ** 2).mean().clamp_min(1e-20))), 1),
                f"{args.precision}_rays_over_1e-3": [{"ray": int(sel[i]), f"{args.precision}_err": float(f"{float(e_pix[i]):.3e}"), "fp32_err": float(f"{float(e32[i]):.3e}"),
                                                      f"rounds_oracle_{args.precision}_fp32": [float(ref["iter_usage"][i]), float(g_ex["iter_usage"][0, i]), float(x32["iter_usage"][0, i])]}
                                                     for i in over16[:16]]}
            del m32, f32, r32, d32, x32
        del g_rgb, g_depth, g_ex
        cpu = {"value": round(n_cpu / tc, 1), "unit": "rays/s", "cores": int(cores), "kind": "port", "cpu_model": cpu_model,
               "parity_of_this_run_on_the_sample": parity,
               "sample": f"{n_cpu} rays strided over the same 480x270 frame in one chunk, {N_SAMPLES}+{N_IMPORTANCE} spp, "
                         f"oracle/render.py volsdf_render on torch-CPU fp32, {int(cores)} threads (fastest of {cands} on a 256-ray calibration), {tc:.1f} s",
               "reference_in_survey_container": {"value": 93.0, "unit": "rays/s", "cores": 8, "cpu_model": "Intel Xeon @ 2.10GHz",
                                                 "what": "the reference's own render_fn at 128 spp, timed once in the survey "
                                                         "container (BASELINE.md section 2: 86-107 rays/s); cannot be re-run on the GPU box"}}

    if cpu is not None:
        # BASELINE configs[0] IN FULL: 64 x 64 rays, 32 coarse + 64 fine spp, the CPU plumbing case - the oracle on the
        # host cores next to the HIP renderer on the same rays
        H1 = W1 = 64
        c2w1, K1 = scene.camera(H1, W1)
        o1, d1, _ = rend_util.get_rays(c2w1[None].to(dev), K1[None].to(dev), H1, W1)
        with torch.no_grad():
            t1 = time.perf_counter()
            ref1 = orender.volsdf_render(sd, o1[0].cpu(), d1[0].cpu(), near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=32,
                                         N_importance=N_IMPORTANCE, max_upsample_steps=kw["max_upsample_steps"], chunk=4096)
            tc1 = time.perf_counter() - t1
        kw1 = dict(kw, N_samples=32)
        render_fn(o1, d1, require_nablas=True, calc_normal=True, detailed_output=False, **kw1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rgb1, _, ex1 = render_fn(o1, d1, require_nablas=True, calc_normal=True, detailed_output=True, **kw1)
        torch.cuda.synchronize()
        tg1 = time.perf_counter() - t1

        def parity1(rgb_, usage_):
            e_ = (rgb_[0].cpu() - ref1["rgb"]).abs().max(dim=-1).values
            same_ = usage_[0].cpu() == ref1["iter_usage"]
            conv_ = ref1["iter_usage"] >= 0                      # rays whose error bound converged on the CPU (the others end on a bisected beta+)
            st_ = conv_ & same_                                  # ... in the same number of rounds on the GPU: the rays the tests hold to a HARD 1e-3
            return {"same_upsampling_rounds_frac": round(float(same_.float().mean()), 5), "rays_over_1e-3": int((e_ > 1e-3).sum()),
                    "rays_over_1e-3_among_converged_same_rounds": int((e_[st_] > 1e-3).sum()),
                    "max_abs_rgb_converged_same_rounds": float(f"{float(e_[st_].max()) if st_.any() else 0.0:.3e}"),
                    "rays_over_1e-3_among_converged": int((e_[conv_] > 1e-3).sum()), "max_abs_rgb_all": float(f"{float(e_.max()):.3e}"),
                    "max_abs_rgb_converged": float(f"{float(e_[conv_].max()) if conv_.any() else 0.0:.3e}"),
                    "psnr_db": round(float(-10 * torch.log10(((rgb_[0].cpu() - ref1["rgb"]) ** 2).mean().clamp_min(1e-20))), 1)}
        par1 = {"never_converged_rays_oracle": int((ref1["iter_usage"] < 0).sum()), args.precision: parity1(rgb1, ex1["iter_usage"])}
        if headline_split and not args.no_secondary:
            m32, _, f32 = scene.build_model("VolSDF", seed=0, beta=args.beta, device=dev, precision="fp32")
            with torch.no_grad():
                r32, _, x32 = f32(o1, d1, require_nablas=True, calc_normal=True, detailed_output=True, **kw1)
            par1["fp32"] = parity1(r32, x32["iter_usage"])
            del m32, f32, r32, x32
        cpu["cfg1_64x64_32spp_in_full"] = {"value": round(H1 * W1 / tc1, 1), "unit": "rays/s", "cores": int(cores), "seconds": round(tc1, 2),
                                           "hip_same_rays": {"value": round(H1 * W1 / tg1, 1), "unit": "rays/s", "ms": round(tg1 * 1e3, 2)},
                                           "parity_all_4096_rays": par1,
                                           "reference_in_survey_container": {"value": 278.0, "unit": "rays/s", "cores": 8}}

    if rank == 0:
        out = {
            "metric": "rays/sec at %dx%dx128spp VolSDF render" % (H, W),
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (functional run, not a measurement)")) if dist is not None else None,
            "devices": devices, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong" if primary_tiles else "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (f32 split into 2 bf16 terms, f32 accumulate)",
                                           "mixed": "pixels bf16x3 (sdf, nabla, radiance, compositing of the 192 final samples: f32 split into 2 bf16 terms, 3 MFMAs per "
                                                    "product, f32 accumulate), Algorithm-1 sampler " +
                                                    ("fp16x1 on error-compensated one-term weights (fp16 act x fp16 weight, 1 MFMA per product, f32 accumulate; "
                                                     "model.calibrate_sampler())" if calibrated else "fp16x2 (fp16 act x fp16 hi+lo weights, 2 MFMAs per product, f32 accumulate)") +
                                                    ", guarded: marginal decisions, never-converged rays and rays still active after round 3 re-sampled in bf16x3",
                                           "fp16x2": "fp16x2 (1 fp16 activation term x 2 fp16 weight terms, f32 accumulate) - EXPERIMENT, not the benchmark precision"}[args.precision], "data": "synthetic",
            "config": {"workload": ("configs[1]" if (H, W) == (480, 270) else "configs[4] frame size" if (H, W) == (960, 540) else "custom frame") +
                                   ": volsdf_fangzhou_nature.yaml dims, %dx%d rays/frame, 128 coarse + 64 fine " % (H, W) +
                                   "spp, pure renderer (no CLIP), synthetic random-weight scene beta=%g" % args.beta,
                       "rays_per_step_per_gpu": H * W, "samples_per_ray": N_SAMPLES + N_IMPORTANCE,
                       "parallelism": ("1 GPU" if world == 1 else
                                       f"one frame per step in 2,048-ray tiles round-robin over {world} ranks, one all_gather" if primary_tiles else
                                       f"views round-robin over {world} rank(s), all_gather of [rays,7] tiles"),
                       "rayschunk": "render_kwargs_test['rayschunk'] = val_rayschunk = %s is PASSED to render_fn, as the reference does (train.py:189, render.py:527); "
                                    "the library reads it as the memory hint it is and launches max(rayschunk, volsdf.DEFAULT_RAYSCHUNK = 131,072) rays "
                                    "(volsdf.launch_rays; results are bit-identical for any chunking: tests/test_gpu_configs.py)" % rk.get("rayschunk"),
                       "sampler": None if args.precision != "mixed" else {
                           "mode": model.mode, "precision": model.sampler_precision, "calibration_s": None if t_cal is None else round(t_cal, 2),
                           "dropped_product_rms_nearest_to_compensated": None if not calibrated else
                               {str(k): [float(f"{a:.2e}"), float(f"{b:.2e}")] for k, (a, b) in getattr(model, "calibration_stats", {}).items()},
                           "what": "model.calibrate_sampler() - the line INTEGRATION.md section A adds to render.py after load_state_dict: Algorithm 1's SDF queries on the "
                                   "1-MFMA kernel (C-ABI precision 5) over one-term fp16 weights whose rounding is error-compensated against this checkpoint's own "
                                   "activations (once per set of weights; `calibration_s`, outside the timed steps like the checkpoint load); `--no-calibrate` / "
                                   "secondary.mixed_as_get_model_ships: the mode as get_model returns it (2-MFMA sampler: what a training step runs)"},
                       "sampler_guard": None if not getattr(model, "sampler_guard", 0.0) else {
                           "guard": model.sampler_guard, "late_round": model.sampler_late_round, "rays_sampled_twice_frac": round(model.render_stats.get("escalated", 0) / max(model.render_stats.get("rays", 1), 1), 5),
                           "second_run_sdf_ms_per_step": round(prof["k_sdf_only_escalation"][0] / args.steps, 3),
                           "second_run_sdf_points_per_step": int(prof["k_sdf_only_escalation"][2] / args.steps),
                           "what": "mixed mode: rays whose convergence decision in Algorithm 1 lies within guard * eps of eps, rays that never converge and rays still "
                                   "active after up-sampling round `late_round` are sampled again on the split-bf16 kernels (nerfart_volsdf_fine_sample_guarded2); share "
                                   "over every render call of this run"},
                       "samples_per_sec": round(value * (N_SAMPLES + N_IMPORTANCE), 1),
                       "iter_usage_hist": hist,
                       "algorithmic_tflop_per_frame": round(flops_frame / 1e12, 2),
                       "end_to_end_tflops": round(flops_frame * (1 if primary_tiles else world) * args.steps / dt / 1e12, 2),
                       "mlp_kernel_ms_per_step": kernels_ms,
                       "ref_3090_rays_per_s": 6480},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "secondary": secondary or None,
        }
        if world > 1:
            n1_name, n1_line = cached_n1_line()
            out.update(scale_fields(world, primary_tiles, value, dt / args.steps * 1e3, secondary,
                                    rank_times[(step_tiles if primary_tiles else step).__name__], n1_name, n1_line))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
