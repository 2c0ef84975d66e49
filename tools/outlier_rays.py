#!/usr/bin/env python
"""Per-ray view of the outliers of one orbit view: every ray of the 2,048-ray oracle sample that ANY sampler arithmetic puts past --thr, with its error and
its up-sampling rounds under each arithmetic (is an outlier the arithmetic's, or a ray every arithmetic moves?).

    python tools/outlier_rays.py --pose 0 [--samplers fp16x2:0.005,fp16x1c:0.005,fp16x1:0.05] 
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pose", type=int, default=0)
    ap.add_argument("--samplers", default="fp16x2:0.005,fp16x1c:0.005,fp16x1:0.05")
    ap.add_argument("--thr", type=float, default=7e-4)
    ap.add_argument("--oracle", default=os.path.join(ROOT, "tests", "golden", "oracle_views_golden.npz"), help="the committed oracle renderings (make_oracle_views.py)")
    args = ap.parse_args()
    from nerfart_amd import scene, rend_util
    dev = "cuda:0"
    H, W, n = 480, 270, 2048
    cache = np.load(args.oracle)
    ref_rgb, ref_use = torch.from_numpy(cache[f"pose{args.pose}_rgb"]), torch.from_numpy(cache[f"pose{args.pose}_iter_usage"])
    c2w, K = scene.camera(H, W, angle=scene.spiral(90)[args.pose])
    o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
    sel = torch.arange(0, H * W, (H * W) // n)[:n]
    model, rk, fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision="bf16x3")
    kw = dict(rk, N_samples=128)
    res = {}
    modes = [("bf16x3", None, 0.0)] + [(f"{s}@{g}", s, float(g)) for s, g in (x.split(":") for x in args.samplers.split(","))]
    for name, samp, guard in modes:
        model.set_sampler_precision(samp, guard=guard) if samp else model.set_sampler_precision(None)
        rgb, _, ex = fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
        res[name] = ((rgb[0].cpu()[sel] - ref_rgb).abs().max(dim=-1).values, ex["iter_usage"][0].cpu()[sel])
        del ex
    bad = torch.zeros(n, dtype=torch.bool)
    for e, u in res.values():
        bad |= e > args.thr
    out = {"pose": args.pose, "thr": args.thr, "rays": []}
    for i in torch.nonzero(bad).flatten().tolist():
        out["rays"].append({"ray": int(sel[i]), "oracle_rounds": int(ref_use[i]), **{m: {"err": float(f"{float(e[i]):.3e}"), "rounds": int(u[i])} for m, (e, u) in res.items()}})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
