#!/usr/bin/env python
"""The guard band of the `mixed` mode's sampler, measured: for guard in --guards and every view in --poses (orbit poses of
scene.spiral(90); bench.py samples pose `warmup`, i.e. 5 on the driver's command and 1 at the defaults):

  * against the pure split-bf16 frame (all 129,600 rays): share of rays whose up-sampling took the same rounds, rays past 1e-3, max;
  * against the CPU oracle on --rays strided rays of the same view: identical rounds, rays past 1e-3 (all / among the rays that converged in the
    same rounds), max, PSNR - the statistics tests/test_gpu_configs.py::pixel_budget holds the shipped mode to;
  * the share of rays the guard sent through Algorithm 1 a second time, and ms per frame over --frames frames of other poses.

    python tools/guard_sweep.py [--guards 0,0.02,0.05,0.1] [--poses 0,1,5,...] [--rays 2048] > profiles/rNN_guard_sweep.json
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def stats(got, ref):
    err = (got - ref).abs().max(dim=-1).values
    n = err.numel()
    return err, {"rays": n, "rays_over_1e-3": int((err > 1e-3).sum()), "max_abs": float(f"{float(err.max()):.3e}"),
                 "p999_abs": float(f"{float(err.flatten().kthvalue(max(1, int(0.999 * n))).values):.3e}"),
                 "psnr_db": round(float(-10 * torch.log10(((got - ref) ** 2).mean().clamp_min(1e-20))), 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--guards", default="0,0.01,0.02,0.05,0.1,0.2")
    ap.add_argument("--poses", default="0,1,5,11,23,37,53,71")
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--size", default="480x270")
    ap.add_argument("--n-samples", type=int, default=128)
    ap.add_argument("--sampler", default="fp16x2", help="the sampler's arithmetic: fp16x2 (2 MFMAs per product, C-ABI precision 4) or fp16x1 (1 MFMA, precision 5)")
    ap.add_argument("--seed", type=int, default=0, help="seed of the synthetic scene's weights (scene.build_model): another member of the scene family")
    ap.add_argument("--beta", type=float, default=0.01, help="the scene's beta (0.01: the benchmark scene; 0.002: the sharper surface of the G9 goldens)")
    ap.add_argument("--late", type=int, default=0, help="late_round of the guarded sampler: rays still active after that round are escalated too")
    ap.add_argument("--oracle-cache", default=None, help="npz of oracle outputs per (size, spp, pose, rays): read if present, written back (the oracle "
                    "costs ~30 s per view on the GPU box's host cores)")
    args = ap.parse_args()
    from nerfart_amd import scene, rend_util, hip
    from oracle import render as orender
    dev = "cuda:0"
    H, W = (int(v) for v in args.size.split("x"))
    guards = [float(g) for g in args.guards.split(",")]
    poses = [int(p) for p in args.poses.split(",")]
    angles = scene.spiral(90)
    mb, rk, fb = scene.build_model("VolSDF", seed=args.seed, beta=args.beta, device=dev, precision="bf16x3")
    mm, _, fm = scene.build_model("VolSDF", seed=args.seed, beta=args.beta, device=dev, precision="bf16x3")
    kw = dict({k: v for k, v in rk.items() if k != "rayschunk"}, N_samples=args.n_samples)
    sd = {k: v.detach().cpu() for k, v in mb.state_dict().items()}
    out = {"frame": f"{H}x{W}, {args.n_samples} + 64 spp, beta {args.beta:g}, scene seed {args.seed}", "oracle_rays": args.rays, "sampler": args.sampler, "late_round": args.late, "csrc_sha256": hip.csrc_sha256(), "views": {}, "timing": {}}

    def frame(fn, o, d):
        rgb, _, ex = fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
        r = (rgb[0].cpu(), ex["iter_usage"][0].cpu())
        del ex
        return r

    def ms_per_frame(fn):
        fn(*rays_t[0], require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o2, d2 in rays_t[1:]:
            fn(o2, d2, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / (len(rays_t) - 1) * 1e3, 2)

    rays_t = []
    for s in range(args.frames + 1):
        c2, K2 = scene.camera(H, W, angle=angles[(7 * s + 3) % 90])
        o2, d2, _ = rend_util.get_rays(c2[None].to(dev), K2[None].to(dev), H, W)
        rays_t.append((o2, d2))
    out["timing"]["bf16x3"] = {"ms_per_frame": ms_per_frame(fb)}
    for g in guards:
        mm.set_sampler_precision(args.sampler, guard=g, late_round=args.late)
        mm.render_stats = {}
        ms = ms_per_frame(fm)
        out["timing"][f"guard_{g:g}"] = {"ms_per_frame": ms, "escalated_frac": round(mm.render_stats["escalated"] / max(mm.render_stats["rays"], 1), 5)}
        print(f"guard {g:g}: {ms} ms / frame, escalated {out['timing'][f'guard_{g:g}']['escalated_frac']}", file=sys.stderr, flush=True)
    import numpy as np
    cache = dict(np.load(args.oracle_cache)) if args.oracle_cache and os.path.exists(args.oracle_cache) else {}
    for pose in poses:
        c2w, K = scene.camera(H, W, angle=angles[pose])
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        n = min(args.rays, H * W)
        sel = torch.arange(0, H * W, (H * W) // n)[:n]
        key = f"{H}x{W}_{args.n_samples}_{pose}_{n}" + ("" if args.beta == 0.01 else f"_beta{args.beta:g}") + ("" if args.seed == 0 else f"_seed{args.seed}")
        if key + "_rgb" in cache:
            ref, t_or = {"rgb": torch.from_numpy(cache[key + "_rgb"]), "iter_usage": torch.from_numpy(cache[key + "_iter_usage"])}, 0.0
        else:
            with torch.no_grad():
                t0 = time.perf_counter()
                ref = orender.volsdf_render(sd, o[0, sel].cpu(), d[0, sel].cpu(), near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=args.n_samples,
                                            max_upsample_steps=kw["max_upsample_steps"], chunk=n)
                t_or = time.perf_counter() - t0
            cache[key + "_rgb"], cache[key + "_iter_usage"] = ref["rgb"].numpy(), ref["iter_usage"].numpy()
            if args.oracle_cache:
                np.savez_compressed(args.oracle_cache, **cache)
        conv = ref["iter_usage"] >= 0
        rec = {"oracle_s": round(t_or, 1), "oracle_never_converged": int((~conv).sum()), "modes": {}}
        base_rgb, base_use = frame(fb, o, d)

        def vs_oracle(rgb, use):
            err, st = stats(rgb[sel], ref["rgb"])
            same = use[sel] == ref["iter_usage"]
            st["same_rounds_frac"] = round(float(same.float().mean()), 5)
            st["over_1e-3_among_converged_same_rounds"] = int((err[same & conv] > 1e-3).sum())
            st["over_1e-3_among_oracle_converged"] = int((err[conv] > 1e-3).sum())
            return st
        rec["modes"]["bf16x3"] = {"vs_oracle": vs_oracle(base_rgb, base_use)}
        for g in guards:
            mm.set_sampler_precision(args.sampler, guard=g, late_round=args.late)
            mm.render_stats = {}
            rgb, use = frame(fm, o, d)
            _, stb = stats(rgb, base_rgb)
            stb["same_rounds_frac"] = round(float((use == base_use).float().mean()), 5)
            stb["rounds_differ"] = int((use != base_use).sum())
            rec["modes"][f"guard_{g:g}"] = {"escalated_frac": round(mm.render_stats["escalated"] / max(mm.render_stats["rays"], 1), 5),
                                            "vs_bf16x3_full_frame": stb, "vs_oracle": vs_oracle(rgb, use)}
        out["views"][f"pose_{pose}"] = rec
        print(f"pose {pose}: " + json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
