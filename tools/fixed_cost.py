#!/usr/bin/env python
"""Fixed (ray-count independent) time of a render call: frames of n rays strided over the 480 x 270 view, ms per call and the
straight-line fit t = a + b n.  What an 8-rank strong-scaling step pays per frame besides its 1/8 of the rays."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerfart_amd import scene, rend_util

dev = "cuda:0"
model, rk, render_fn = scene.build_model("VolSDF", seed=0, beta=0.01, device=dev, precision="bf16x3")
kw = {k: v for k, v in rk.items() if k != "rayschunk"}
H, W = 480, 270
c2w, K = scene.camera(H, W)
o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
res = {}
for n in (256, 2048, 8100, 16200, 32400, 64800, 129600):
    # interleaved 2,048-ray tiles as dist.render_sharded deals them (n / 2048 tiles spread over the frame)
    tiles = torch.arange(0, H * W, 2048)
    take = tiles[:: max(1, len(tiles) * 2048 // n)][: max(1, n // 2048)]
    idx = torch.cat([torch.arange(int(s), min(int(s) + min(2048, n), H * W)) for s in take])[:n].to(dev)
    ro, rd = o[:, idx].contiguous(), d[:, idx].contiguous()
    for _ in range(2):
        render_fn(ro, rd, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        render_fn(ro, rd, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
    torch.cuda.synchronize()
    res[int(idx.numel())] = round((time.perf_counter() - t0) / reps * 1e3, 3)
ns = sorted(res)
import numpy as np
A = np.stack([np.ones(len(ns)), np.array(ns, dtype=float)], 1)
a, b = np.linalg.lstsq(A[2:], np.array([res[n] for n in ns])[2:], rcond=None)[0]
print(json.dumps({"ms_per_call": res, "fit_ms": {"fixed": round(float(a), 3), "per_1000_rays": round(float(b) * 1e3, 4)}}))
