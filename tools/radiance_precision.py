#!/usr/bin/env python
"""The survey's cheap-arithmetic experiment on the radiance net (VERDICT r05 next 3 i): the SAME sampling and the same sdf / nabla
(split-bf16), only the radiance MLP of the 192 final samples at another C-ABI precision (nerfart_volsdf_render_staged_fwd's rad_precision):
per pose the pixel difference against the split-bf16 radiance frame (all rays: identical samples, so this is the radiance arithmetic alone),
against the CPU oracle on the strided sample, and ms per frame.

    python tools/radiance_precision.py [--precisions fp16x2] [--poses 0,5,23] > profiles/rNN_radiance_precision.json
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precisions", default="fp16x2")
    ap.add_argument("--poses", default="0,5,23")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--mode", default="mixed")
    args = ap.parse_args()
    from nerfart_amd import scene, rend_util, hip
    dev = "cuda:0"
    H, W = 480, 270
    angles = scene.spiral(90)
    model, rk, fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision=args.mode)
    kw = {k: v for k, v in rk.items() if k != "rayschunk"}
    out = {"frame": f"{H}x{W}", "mode": args.mode, "csrc_sha256": hip.csrc_sha256(), "precisions": {}}
    rays_t = []
    for s in range(args.frames + 1):
        c2, K2 = scene.camera(H, W, angle=angles[(7 * s + 3) % 90])
        o2, d2, _ = rend_util.get_rays(c2[None].to(dev), K2[None].to(dev), H, W)
        rays_t.append((o2, d2))

    def ms_per_frame():
        fn(*rays_t[0], require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o2, d2 in rays_t[1:]:
            fn(o2, d2, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / (len(rays_t) - 1) * 1e3, 2)

    views = []
    for pose in [int(p) for p in args.poses.split(",")]:
        c2w, K = scene.camera(H, W, angle=angles[pose])
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        views.append((pose, o, d))
    model.set_radiance_precision(None)
    base = {pose: fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw) for pose, o, d in views}
    base = {p: (r[0][0].clone(), r[2]["radiance"][0].clone(), r[2]["d_vals"][0].clone()) for p, r in base.items()}
    hip.profile_begin()
    out["precisions"]["model"] = {"ms_per_frame": ms_per_frame()}
    prof = hip.profile_end()
    out["precisions"]["model"]["k_radiance_ms_per_frame"] = round(prof["k_radiance"][0] / args.frames, 3)
    for prec in args.precisions.split(","):
        model.set_radiance_precision(prec)
        rec = {"views": {}}
        for pose, o, d in views:
            rgb, _, ex = fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
            assert torch.equal(ex["d_vals"][0], base[pose][2]), "the sampling must not depend on the radiance precision"
            e = (rgb[0] - base[pose][0]).abs().max(dim=-1).values
            er = (ex["radiance"][0] - base[pose][1]).abs()
            rec["views"][f"pose_{pose}"] = {"pixel_max_abs": float(f"{float(e.max()):.3e}"), "pixel_p999": float(f"{float(e.flatten().kthvalue(int(0.999 * e.numel())).values):.3e}"),
                                            "pixel_rays_over_1e-4": int((e > 1e-4).sum()), "pixel_rays_over_1e-3": int((e > 1e-3).sum()),
                                            "per_sample_radiance_max_abs": float(f"{float(er.max()):.3e}"), "per_sample_radiance_rms": float(f"{float((er ** 2).mean().sqrt()):.3e}")}
        hip.profile_begin()
        rec["ms_per_frame"] = ms_per_frame()
        prof = hip.profile_end()
        rec["k_radiance_ms_per_frame"] = round(prof["k_radiance"][0] / args.frames, 3)
        out["precisions"][prec] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
