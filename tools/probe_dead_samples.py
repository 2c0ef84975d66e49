#!/usr/bin/env python
"""How many of the 192 final samples of a ray lie behind the point where the transmittance has underflowed (their weights are 0:
the renderer's rgb / depth / normals do not depend on them)?  Benchmark scene, 480x270, strided rays; segment-granular fractions."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfart_amd import scene, rend_util

dev = torch.device("cuda", 0)
out = {}
for beta in (0.01, 0.002, 0.1):
    model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=beta, device=dev, precision="bf16x3")
    H, W = 480, 270
    c2w, K = scene.camera(H, W)
    o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
    sel = torch.arange(0, H * W, 7, device=dev)
    kw = {k: v for k, v in rk.items() if k != "rayschunk"}
    rgb, depth, ex = render_fn(o[:, sel], d[:, sel], require_nablas=True, calc_normal=True, detailed_output=True, **kw)
    sigma = ex["sigma"][0].double()
    dv = ex["d_vals"][0].double()
    delta = dv[:, 1:] - dv[:, :-1]
    x = torch.relu(sigma[:, :-1] * delta)
    logT = -torch.cumsum(x, dim=1)                                  # log T after sample i
    logT = torch.cat([torch.zeros_like(logT[:, :1]), logT], dim=1)  # log T_i (before sample i)
    P = dv.shape[1]
    res = {}
    for thr, name in ((-103.0, "T_underflows_fp32"), (-69.0, "T_below_1e-30")):
        dead = logT < thr
        res[name] = {"points_dead_frac": round(float(dead.float().mean()), 4)}
        for seg in (16, 24, 32, 48, 64):
            ns = P // seg
            dseg = dead[:, :ns * seg].reshape(-1, ns, seg).all(dim=2)     # whole segment dead
            res[name][f"segments_of_{seg}_dead_frac"] = round(float(dseg.float().mean()), 4)
    res["rays_hitting"] = round(float((logT[:, -1] < -69).float().mean()), 4)
    out[f"beta_{beta}"] = res
print(json.dumps(out))
