#!/usr/bin/env python
"""One table for every precision mode of the renderer: the statistics the shipped mode is held to -
rays past 1e-3, max, p99.9, PSNR - against the CPU oracle on 2,048 strided rays and against the exact-fp32 HIP frame on all 129,600
rays, on the default camera and on bench.py's view (orbit pose 1), plus ms per frame (3 frames).  Every ray a mode puts past 1e-3
against the oracle is listed with the OTHER modes' errors on the same ray (is it the arithmetic, or one of Algorithm 1's
discontinuities that flips under any rounding?).

    python tools/parity_table.py [--precisions fp32,calibrated,mixed,bf16x3,fp16x2] [--rays 2048] > profiles/rNN_parity_table.json
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def stats(got, ref):
    err = (got - ref).abs().max(dim=-1).values
    n = err.numel()
    return err, {"rays": n, "rays_over_1e-3": int((err > 1e-3).sum()), "over_frac": round(float((err > 1e-3).float().mean()), 6),
                 "max_abs": float(f"{float(err.max()):.3e}"), "p999_abs": float(f"{float(err.flatten().kthvalue(max(1, int(0.999 * n))).values):.3e}"),
                 "psnr_db": round(float(-10 * torch.log10(((got - ref) ** 2).mean().clamp_min(1e-20))), 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precisions", default="fp32,calibrated,mixed,bf16x3,fp16x2")
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=3)
    args = ap.parse_args()
    from nerfart_amd import scene, rend_util, hip
    from oracle import render as orender
    dev = "cuda:0"
    H, W = 480, 270
    MIXED = "mixed"            # the shipped mode: Algorithm 1 on the 2-MFMA kernels, the 192 final samples in split-bf16 (set_precision("mixed"))
    CAL = "calibrated"         # ... and its rendering form: mixed + model.calibrate_sampler() (the 1-MFMA sampler on error-compensated one-term weights)
    precisions = [p for p in args.precisions.split(",") if p in hip.PRECISIONS or p in (MIXED, CAL)]
    if "fp32" not in precisions:
        precisions = ["fp32"] + precisions
    angles = scene.spiral(90)
    out = {"frame": f"{H}x{W}, 128 + 64 spp, beta 0.01", "oracle_rays": args.rays, "csrc_sha256": hip.csrc_sha256(), "views": {}}
    models = {p: scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision=MIXED if p == CAL else p) for p in precisions}
    if CAL in models:
        models[CAL][0].calibrate_sampler()
    sd = {k: v.detach().cpu() for k, v in models["fp32"][0].state_dict().items()}
    for view, ang in (("default", 0.0), ("bench_orbit_pose_1", angles[1])):
        c2w, K = scene.camera(H, W, angle=ang)
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        sel = torch.arange(0, H * W, (H * W) // args.rays)[:args.rays]
        with torch.no_grad():
            t0 = time.perf_counter()
            ref = orender.volsdf_render(sd, o[0, sel].cpu(), d[0, sel].cpu(), near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=128, max_upsample_steps=6,
                                        chunk=args.rays)
            t_or = time.perf_counter() - t0
        frames, usage, rec = {}, {}, {"oracle_s": round(t_or, 1), "modes": {}}
        for p in precisions:
            model, rk, fn = models[p]
            kw = {k: v for k, v in rk.items() if k != "rayschunk"}
            rgb, _, ex = fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
            frames[p], usage[p] = rgb[0].cpu(), ex["iter_usage"][0].cpu()
            del ex
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(args.frames):
                c2, K2 = scene.camera(H, W, angle=angles[(7 * s + 3) % 90])
                o2, d2, _ = rend_util.get_rays(c2[None].to(dev), K2[None].to(dev), H, W)
                fn(o2, d2, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
            torch.cuda.synchronize()
            rec["modes"][p] = {"ms_per_frame": round((time.perf_counter() - t0) / args.frames * 1e3, 1)}
        errs = {}
        for p in precisions:
            errs[p], st = stats(frames[p][sel], ref["rgb"])
            st["same_upsampling_rounds_frac"] = round(float((usage[p][sel] == ref["iter_usage"]).float().mean()), 5)
            rec["modes"][p]["vs_oracle"] = st
            if p != "fp32":
                _, st32 = stats(frames[p], frames["fp32"])
                st32["same_upsampling_rounds_frac"] = round(float((usage[p] == usage["fp32"]).float().mean()), 5)
                rec["modes"][p]["vs_hip_fp32_full_frame"] = st32
        outl = sorted({int(i) for p in precisions for i in (errs[p] > 1e-3).nonzero().flatten().tolist()})
        rec["rays_past_1e-3_vs_oracle"] = [
            {"ray": int(sel[i]), "rounds_oracle": float(ref["iter_usage"][i]),
             **{p: {"err": float(f"{float(errs[p][i]):.3e}"), "rounds": float(usage[p][sel][i])} for p in precisions}} for i in outl]
        out["views"][view] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
