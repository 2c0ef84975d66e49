#!/usr/bin/env python
"""CPU estimate of what a cheaper Algorithm-1 sampler costs in pixels BEFORE a kernel is written for it (round 6).

The `mixed` mode runs only the sampler's SDF queries (512 (1 + rounds) per ray, no gradient, volsdf.py:479) on cheap arithmetic; every value that
reaches a pixel is computed afterwards at the model's precision.  This script plays that on the CPU oracle: `oracle.render.volsdf_render` with the
sampler's `volsdf_forward_surface` replaced by a model of the kernel's arithmetic -

    x2: hidden activations rounded to fp16 (one term), weights fp16 hi + lo  (csrc/mlp_chain_f16x2.hip, C-ABI precision 4: 2 MFMAs per product)
    x1: hidden activations rounded to fp16, weights rounded to fp16 (one term each: 1 MFMA per product)
    b1: the same in bf16 (8 bits)

(the ready-made input units - the positional encodings of layers 0 and 4 - keep hi + lo terms in every variant, as in the kernel; accumulation in
fp32) - and counts, against the untouched oracle on the same rays: rays past 1e-3, the same among the rays with identical rounds, max, PSNR, identical
rounds; and the same after a GUARD: rays whose max B came within guard * eps of eps at any convergence check, and rays that never converged, take
the exact sampler's samples (what nerfart_volsdf_fine_sample_guarded does with the split-bf16 kernels).

    python tools/emul_sampler_precision.py [--pose 1] [--rays 512] [--guards 0,0.005,0.02,0.05]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import torch.nn.functional as F


def q(x, dt):
    return x.to(dt).float()


def surface_q(sd, x, variant, multires=6, skips=(4,), stem="implicit_surface.surface_fc_layers"):
    from oracle import nets
    dt = torch.bfloat16 if variant == "b1" else torch.float16
    e = nets.embed(x, multires)
    h = e
    D = nets.n_layers(sd, stem) - 1
    for i in range(D):
        W, b = nets.folded_weight(sd, f"{stem}.{i}"), sd[f"{stem}.{i}.bias"]
        if i == 0:
            z = F.linear(e, W, b)                                   # ready-made units: three-term form
        else:
            if i in skips:
                W = W / np.sqrt(2)
                nh = h.shape[-1]
                Wh, We = W[:, :nh], W[:, nh:]
            else:
                Wh, We = W, None
            Whq = q(Wh, dt) if variant in ("x1", "b1") else q(Wh, dt) + q(Wh - q(Wh, dt), dt)
            z = F.linear(q(h, dt), Whq, b)
            if We is not None:
                z = z + F.linear(e, We)
        h = nets.softplus100(z)
    W, b = nets.folded_weight(sd, f"{stem}.{D}"), sd[f"{stem}.{D}.bias"]
    return F.linear(h, W[:1], b[:1])[..., 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pose", type=int, default=1)
    ap.add_argument("--rays", type=int, default=512)
    ap.add_argument("--variants", default="x2,x1")
    ap.add_argument("--guards", default="0,0.005,0.02,0.05,0.1")
    args = ap.parse_args()
    import make_oracle_views as mov
    from oracle import nets, sampling
    from oracle import render as orender
    sd = mov.scene_sd()
    idx = torch.arange(0, mov.N, mov.N // args.rays)[:args.rays]
    _, o, d = mov.view_rays(args.pose, idx)
    eps = 0.1
    margins = {}

    # instrument: the convergence checks are the error_bound calls with the net's (scalar) alpha
    eb0 = sampling.error_bound
    state = {"rows": None, "margin": None}

    def render(variant):
        fs0 = orender.fine_sample

        def fs(sdf_fn, d_init, rays_o, rays_d, alpha_net, beta_net, far, **kw):
            R = d_init.shape[0]
            margin = torch.full((R,), float("inf"))
            live = {"idx": torch.arange(R)}

            def eb(dv, sv, alpha, beta):
                out = eb0(dv, sv, alpha, beta)
                if alpha is alpha_net:                                                   # a check at the net's beta
                    mx = out.max(dim=-1).values
                    if mx.shape[0] != live["idx"].shape[0]:
                        raise RuntimeError("active set tracking lost")
                    m = (mx - eps).abs() / eps
                    margin[live["idx"]] = torch.minimum(margin[live["idx"]], m)
                    live["idx"] = live["idx"][mx > eps]
                return out
            sampling.error_bound = eb
            try:
                r = fs0(sdf_fn, d_init, rays_o, rays_d, alpha_net, beta_net, far, **kw)
            finally:
                sampling.error_bound = eb0
            state["margin"] = margin
            return r
        orender.fine_sample = fs
        vfs0 = nets.volsdf_forward_surface
        if variant != "exact":
            def vfs(sd_, x, R=3.0, multires=6, skips=(4,)):
                s = surface_q(sd_, x, variant, multires, skips)
                return torch.min(s, R - x.norm(dim=-1)), None
            nets.volsdf_forward_surface = vfs
        try:
            with torch.no_grad():
                ret = orender.volsdf_render(sd, o, d, near=0.0, far=6.0, obj_bounding_radius=3.0, N_samples=128, max_upsample_steps=6, chunk=max(o.shape[0], 64))
        finally:
            nets.volsdf_forward_surface = vfs0
            orender.fine_sample = fs0
        return ret["rgb"], ret["iter_usage"], state["margin"].clone()

    ref_rgb, ref_it, ref_margin = render("exact")
    out = {"pose": args.pose, "rays": int(o.shape[0]), "never_converged_exact": int((ref_it < 0).sum()), "variants": {}}
    # sdf error of the variants on the first round's points
    pts = (o[:, None, :] + F.normalize(d, dim=-1)[:, None, :] * torch.linspace(0, 6, 64)[None, :, None]).reshape(-1, 3)
    with torch.no_grad():
        s_exact = nets.surface_forward(sd, pts)[0]
    for v in args.variants.split(","):
        with torch.no_grad():
            sv = surface_q(sd, pts, v)
        rgb, it, margin = render(v)
        res = {"sdf_err_max": float((sv - s_exact).abs().max()), "sdf_err_mean": float((sv - s_exact).abs().mean()), "guards": {}}
        for g in (float(x) for x in args.guards.split(",")):
            esc = (it < 0) | (margin <= g) if g > 0 else torch.zeros_like(it, dtype=torch.bool)
            got = torch.where(esc[:, None], ref_rgb, rgb)
            git = torch.where(esc, ref_it, it)
            err = (got - ref_rgb).abs().max(dim=-1).values
            same = git == ref_it
            res["guards"][f"{g:g}"] = {
                "escalated_frac": round(float(esc.float().mean()), 4), "identical_rounds": round(float(same.float().mean()), 4),
                "rays_over_1e-3": int((err > 1e-3).sum()), "same_rounds_over_1e-3": int(((err > 1e-3) & same).sum()),
                "converged_over_1e-3": int(((err > 1e-3) & (ref_it >= 0) & (git >= 0)).sum()),
                "max_abs": float(f"{float(err.max()):.3e}"), "p99_abs": float(f"{float(err.kthvalue(max(1, int(0.99 * err.numel()))).values):.3e}"),
                "psnr_db": round(float(-10 * torch.log10(((got - ref_rgb) ** 2).mean().clamp_min(1e-20))), 1)}
        out["variants"][v] = res
        print(json.dumps({v: res}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
