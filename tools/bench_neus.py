#!/usr/bin/env python
"""NeuS render timing (BASELINE configs[3] / SURVEY cfg4: neus_fangzhou_vangogh.yaml dims, 64 + 64 samples per ray,
480x270) on one MI355X.  Prints one JSON line (NOT the driver's bench contract - that is bench.py)."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--precision", default="bf16x3")
    args = ap.parse_args()
    from nerfart_amd import scene, rend_util
    dev = torch.device("cuda", 0)
    model, rk, render_fn = scene.build_model("NeuS", seed=0, beta=None, device=dev, precision=args.precision)
    H, W = 480, 270
    kw = {k: v for k, v in rk.items() if k != "rayschunk"}
    ts = []
    for it in range(args.steps + 1):
        c2w, K = scene.camera(H, W, angle=0.1 * it)
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rgb, depth, ex = render_fn(o, d, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sum(ts[1:]) / args.steps
    print(json.dumps({"workload": "NeuS render 480x270, 64+64 spp (704.8 MFLOP/ray algorithmic)", "precision": args.precision,
                      "ms_per_frame": round(t * 1e3, 1), "rays_per_s": round(H * W / t, 1),
                      "algorithmic_tflops": round(H * W * 704.8e6 / t / 1e12, 1)}))


if __name__ == "__main__":
    main()
