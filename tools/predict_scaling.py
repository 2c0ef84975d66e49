#!/usr/bin/env python
"""Multi-GPU readiness without a second GPU: predicted scaling of the ray-sharded render from MEASURED per-ray work.

    python tools/predict_scaling.py dump    <maps.npz>        # on the GPU box: iter_usage maps + single-GPU frame times
    python tools/predict_scaling.py predict <maps.npz> [out]  # anywhere: per-rank work for N = 2 / 4 / 8 -> predicted efficiency

`dump` renders, on one MI355X, the 480 x 270 and 960 x 540 frames of the synthetic scene at beta = 0.01 and 0.002 (default camera) and 16
orbit views at 480 x 270 / beta 0.01, and stores every ray's up-sampling rounds (uint8, rounds + 1; 0 = never converged) with the measured
ms per frame.  `predict` prices a ray at SURVEY 8d's algorithmic flops - 512 (1 + u) F_sdf + 192 (F_sdf + F_grad + F_rad), u = rounds (6 if
never converged) -, deals `tile`-ray tiles round-robin exactly as nerfart_amd.dist.tile_assignment does, and models a rank's frame as

    t_rank = t_fixed + (flops_rank / flops_frame) (T1 - t_fixed) + t_allgather(N)

with T1 the measured single-GPU frame time, t_fixed = 0.84 ms (profiles/r03j_fixed_cost.json: the per-call cost that does not shrink with
the ray count) and t_allgather = (N - 1) (latency + shard_bytes / link_rate): a ring all_gather of [rays, 7] fp32 over xGMI, priced at
a deliberately conservative 50 GB/s per link (MI355X_MICROARCH.md: 153.6 GB/s peak per link and direction) and 20 us per step.
Efficiency (strong) = T1 / (N max_rank t_rank).  The weak mode of bench.py (one orbit view per rank per step) is bounded by the slowest
of the N views of a step: efficiency = mean(T_view) / max(T_view) over consecutive groups of N views."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

F_SDF, F_NABLA, F_RAD = 1049088, 918016, 530432
T_FIXED_MS, LINK_GBPS, STEP_LAT_US = 0.84, 50.0, 20.0


def ray_flops(u8):
    r = u8.astype(np.float64) - 1.0
    r[u8 == 0] = 6.0
    return 512.0 * (1.0 + r) * F_SDF + 192.0 * (F_SDF + F_NABLA + F_RAD)


def dump(path):
    import time
    import torch
    from nerfart_amd import scene, rend_util
    dev = "cuda:0"
    out = {}

    def frame(model_rk_fn, H, W, angle, reps):
        _, rk, fn = model_rk_fn
        kw = {k: v for k, v in rk.items() if k != "rayschunk"}
        c2w, K = scene.camera(H, W, angle=angle)
        o, d, _ = rend_util.get_rays(c2w[None].to(dev), K[None].to(dev), H, W)
        _, _, ex = fn(o, d, require_nablas=True, calc_normal=True, detailed_output=True, **kw)
        u = ex["iter_usage"][0]
        del ex
        fn(o, d, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(o, d, require_nablas=True, calc_normal=True, detailed_output=False, **kw)
        torch.cuda.synchronize()
        return (u + 1).clamp_min(0).to(torch.uint8).cpu().numpy(), (time.perf_counter() - t0) / reps * 1e3
    for beta in (0.01, 0.002):
        m = scene.build_model("VolSDF", seed=0, beta=beta, device=dev, precision="bf16x3")
        for (H, W) in ((480, 270), (960, 540)):
            u, ms = frame(m, H, W, 0.0, 3)
            out[f"usage_{H}x{W}_beta{beta}"] = u
            out[f"ms_{H}x{W}_beta{beta}"] = np.float64(ms)
        if beta == 0.01:
            ang = scene.spiral(90)
            us, mss = [], []
            for v in range(16):
                u, ms = frame(m, 480, 270, ang[v], 1)
                us.append(u); mss.append(ms)
            out["usage_orbit16_480x270_beta0.01"] = np.stack(us)
            out["ms_orbit16_480x270_beta0.01"] = np.array(mss)
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items()})


def predict(path, out_path=None):
    z = np.load(path)
    res = {"model": {"t_fixed_ms": T_FIXED_MS, "link_GB_per_s": LINK_GBPS, "allgather_step_latency_us": STEP_LAT_US,
                     "ray_flops": "512 (1 + u) F_sdf + 192 (F_sdf + F_grad + F_rad), u = iter_usage (6 where -1)",
                     "tiles": "tile-ray tiles dealt round-robin (nerfart_amd.dist.tile_assignment)",
                     "formula": "t_rank = t_fixed + flops_rank / flops_frame (T1 - t_fixed) + (N - 1) (latency + shard_bytes / link_rate); efficiency = T1 / (N max t_rank)"},
           "strong_tiles": {}, "weak_views": {}}
    for key in sorted(k for k in z.files if k.startswith("usage_") and "orbit" not in k):
        name = key[len("usage_"):]
        u = z[key]
        T1 = float(z["ms_" + name])
        f = ray_flops(u)
        n = f.size
        rec = {"rays": int(n), "T1_ms_measured": round(T1, 2), "tflop_per_frame": round(f.sum() / 1e12, 2),
               "rounds_hist": {str(int(k) - 1): int(v) for k, v in zip(*np.unique(u, return_counts=True))}, "N": {}}
        for N in (2, 4, 8):
            row = {}
            for tile in (2048, 1024, 512):
                owner = (np.arange(n) // tile) % N
                fr = np.array([f[owner == q].sum() for q in range(N)])
                n_max = max(int((owner == q).sum()) for q in range(N))
                t_ag = (N - 1) * (STEP_LAT_US * 1e-3 + n_max * 7 * 4 / (LINK_GBPS * 1e9) * 1e3)
                t_rank = T_FIXED_MS + fr / f.sum() * (T1 - T_FIXED_MS) + t_ag
                row[str(tile)] = {"efficiency": round(T1 / (N * t_rank.max()), 4), "speedup": round(T1 / t_rank.max(), 2),
                                  "max_over_mean_flops": round(float(fr.max() / fr.mean()), 4), "t_rank_max_ms": round(float(t_rank.max()), 2),
                                  "t_allgather_ms": round(t_ag, 3)}
            rec["N"][str(N)] = row
        res["strong_tiles"][name] = rec
    if "usage_orbit16_480x270_beta0.01" in z.files:
        us, ms = z["usage_orbit16_480x270_beta0.01"], z["ms_orbit16_480x270_beta0.01"]
        fl = np.array([ray_flops(u).sum() for u in us])
        rec = {"views": int(len(ms)), "ms_per_view_measured": [round(float(v), 1) for v in ms], "tflop_per_view": [round(float(v) / 1e12, 1) for v in fl], "N": {}}
        for N in (2, 4, 8):
            groups = [ms[i:i + N] for i in range(0, len(ms) - N + 1, N)]
            t_ag = (N - 1) * (STEP_LAT_US * 1e-3 + us.shape[1] * 7 * 4 / (LINK_GBPS * 1e9) * 1e3)
            eff = [float(g.mean() / (g.max() + t_ag)) for g in groups]
            rec["N"][str(N)] = {"efficiency_mean_over_steps": round(float(np.mean(eff)), 4), "efficiency_worst_step": round(float(np.min(eff)), 4),
                                "t_allgather_ms": round(t_ag, 3)}
        res["weak_views"]["480x270_beta0.01"] = rec
    txt = json.dumps(res, indent=1)
    if out_path:
        open(out_path, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "dump":
        dump(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "predict":
        predict(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        print(__doc__)
