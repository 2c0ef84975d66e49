#!/usr/bin/env python
"""How far do Algorithm 1's discontinuities move pixels under rounding-level noise ALONE (no GPU, no second implementation)?
The CPU oracle renders a frame twice: with the scene's weights, and with every SDF weight_v entry multiplied by (1 + eps u),
u ~ U(-1, 1), eps = 2^-23 (one fp32 ulp: what another GEMM summation order, FMA contraction or libm does to the SDF values).  Rays
whose error-bounded up-sampling never converges (iter_usage -1) end on a bisected beta+, and a 1e-7 change of one SDF value can move
that bisection's branch - the same rays any other arithmetic (the exact-fp32 HIP mode, bf16x3) lands differently on.

    python tools/oracle_sensitivity.py [--cfg 1|2] [--rays N] > profiles/rNN_oracle_sensitivity_cfgK.json
cfg 1: the whole 64 x 64 frame at 32 + 64 spp (BASELINE configs[0]); cfg 2: N strided rays of the 480 x 270 frame at 128 + 64 spp."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=1)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--eps", type=float, default=2.0 ** -23)
    args = ap.parse_args()
    from nerfart_amd import scene
    from oracle import render as orender
    from conftest import scene_state
    sd, _ = scene_state("VolSDF", 0.01)
    if args.cfg == 1:
        H = W = 64; ns = 32
        sel = torch.arange(H * W)
    else:
        H, W, ns = 480, 270, 128
        sel = torch.arange(0, H * W, (H * W) // args.rays)[:args.rays]
    c2w, K = scene.camera(H, W)
    o, d = orender.get_rays(c2w, K, H, W)[:2]
    o, d = o[sel], d[sel]
    g = torch.Generator().manual_seed(0)
    sd2 = {k: (v * (1.0 + args.eps * (2.0 * torch.rand(v.shape, generator=g) - 1.0)) if ("surface_fc_layers" in k and k.endswith("weight_v")) else v.clone())
           for k, v in sd.items()}
    res = []
    for s in (sd, sd2):
        with torch.no_grad():
            res.append(orender.volsdf_render(s, o, d, near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=ns, N_importance=64, max_upsample_steps=6,
                                             chunk=max(int(sel.numel()), 1024)))
    a, b = res
    e = (a["rgb"] - b["rgb"]).abs().max(-1).values
    conv = a["iter_usage"] >= 0
    print(json.dumps({
        "what": f"oracle vs oracle with SDF weight_v * (1 + {args.eps:.3g} U(-1, 1)): cfg {args.cfg}, {int(e.numel())} rays of {H}x{W}, {ns} + 64 spp",
        "rays": int(e.numel()), "never_converged_rays": int((~conv).sum()),
        "rays_over_1e-3": int((e > 1e-3).sum()), "rays_over_1e-3_among_converged": int((e[conv] > 1e-3).sum()), "rays_over_1e-4": int((e > 1e-4).sum()),
        "max_abs_rgb": float(f"{float(e.max()):.3e}"), "max_abs_rgb_converged": float(f"{float(e[conv].max()):.3e}"),
        "psnr_db": round(float(-10 * torch.log10(((a['rgb'] - b['rgb']) ** 2).mean().clamp_min(1e-20))), 1),
        "same_upsampling_rounds_frac": round(float((a["iter_usage"] == b["iter_usage"]).float().mean()), 5),
        "rounds_hist": {str(int(k)): int(v) for k, v in zip(*[t.tolist() for t in torch.unique(a["iter_usage"], return_counts=True)])}}, indent=1))


if __name__ == "__main__":
    main()
